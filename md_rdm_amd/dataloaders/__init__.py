"""Input pipeline: the reference's dataloaders/ with the per-sample PIL work moved to the GPU."""
from .nyu import NYUDataset, NyuGpuPreprocessor, PrefetchLoader, draw_training_params, identity_params, test_params  # noqa: F401
