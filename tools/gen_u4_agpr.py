"""gemm_u4_agpr.inc: the literal AGPR names of accumulator fragment (mi, ni) of gemm_u4_kernel - AR_ (tuple), CL_ (clobber list), AS_ (single registers)."""
for mi in range(8):
    for ni in range(8):
        b = (mi * 8 + ni) * 4
        print(f'#define AR_{mi}_{ni} "a[{b}:{b + 3}]"')
        print(f'#define CL_{mi}_{ni} "a{b}", "a{b + 1}", "a{b + 2}", "a{b + 3}"')
        for r in range(4):
            print(f'#define AS_{mi}_{ni}_{r} "a{b + r}"')
