"""datasets package of the MI355X build.  The reference's KITTI/Cityscapes/NYU loaders are host-side
image I/O and outside the hot path; this build feeds KITTI-shaped synthetic batches
with the same dict schema."""
from .synthetic import SyntheticKITTIDataset, synthetic_batch
