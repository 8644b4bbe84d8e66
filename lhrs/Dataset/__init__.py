"""lhrs.Dataset: loaders, transform, prompt templates of the training / chat path."""
