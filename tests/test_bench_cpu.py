"""Host-side pieces of bench.py that need no GPU: which committed PMC pass the roofline's `traffic` is quoted from, the F_alg formula of SURVEY §8(d), and the
argument defaults the driver relies on (N = 1, a K / W that finish within minutes)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_traffic_is_quoted_from_the_pass_at_the_same_micro_batch_and_kernel():
    dom = bench.KIND_NAMES.index("gemm_u4_kernel<0, false> plain (four waves)")
    for B, name in ((240, "r06_gemm_traffic_b240.json"), (120, "r06_gemm_traffic_b120.json"), (60, "r06_gemm_traffic.json")):
        want = json.load(open(os.path.join(ROOT, "profiles", name)))
        got, note = bench.gemm_traffic(dom, B, 1.0)
        assert got == int(want["traffic_bytes_per_launch"]) and name in note and want["micro_batch"] == B, (B, note)
        assert 2.0 * 1024 * want["fetch_size_kb_mean"] + 1024 * want["write_size_kb_mean"] == got or abs(2048 * want["fetch_size_kb_mean"] + 1024 * want["write_size_kb_mean"] - got) < 2
    # another kernel, another batch or a truncated model: nothing is quoted
    assert bench.gemm_traffic(bench.KIND_NAMES.index("gemm_u4_kernel<0, true> plain + residual (four waves)"), 240, 1.0)[0] is None
    assert bench.gemm_traffic(dom, 15, 1.0)[0] is None and bench.gemm_traffic(dom, 240, 0.5)[0] is None


def test_f_alg_is_the_survey_formula_at_the_headline_shape():
    assert abs(bench.f_alg(273) - (278.8e9 + 2 * 273 * 13.214e9 + 1572864.0 * 273 * 273)) < 1.0
    assert 7.60e12 < bench.f_alg(273) < 7.62e12          # 7.61 TFLOP per sample


def test_defaults_are_one_gpu_and_a_short_run(monkeypatch):
    import argparse
    seen = {}
    real = argparse.ArgumentParser.parse_args

    def spy(self, *a, **k):
        ns = real(self, [])
        seen.update(vars(ns))
        raise SystemExit(0)
    monkeypatch.setattr(argparse.ArgumentParser, "parse_args", spy)
    try:
        bench.main()
    except SystemExit:
        pass
    assert seen["gpus"] == 1 and seen["steps"] <= 20 and seen["warmup"] <= 5 and seen["micro_batch"] == 240 and seen["stage"] == 1 and seen["llama_layers"] == 32
