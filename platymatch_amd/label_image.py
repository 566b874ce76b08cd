"""Label image -> nucleus centroids and sizes: the label-image branch of the widget's worker
(platymatch/_dock_widget.py:497-521) — `np.where(data == id)` + `np.mean` per label — as ONE pass over the volume on the
device (64-bit integer moment sums, exact), and the RANSAC tolerance the widget derives from the sizes (:613-618)."""
import numpy as np

from . import _kernels as K
from . import _native as nat


def label_centroids(label_image, anisotropy=1.0):
    """-> (detections 3 x K float64 rows z, y, x (mean voxel index per label, ascending label id, as np.unique orders
    them), sizes K = anisotropy * voxel count, ids K).  Label 0 is background (:500-502)."""
    torch = nat.torch_mod()
    lab = label_image if nat.is_torch(label_image) else torch.as_tensor(np.ascontiguousarray(label_image))
    if lab.dim() != 3:
        raise ValueError("label image must be 3-D (z, y, x)")
    if lab.dtype.is_floating_point:
        raise ValueError("label image must have an integer dtype")
    lab = lab.to(nat.device(), dtype=torch.int32).contiguous()
    counts, sums = K.label_moments(lab)
    counts_h, sums_h = counts.cpu().numpy(), sums.cpu().numpy()
    ids = np.nonzero(counts_h)[0]                                            # ascending, background excluded
    cnt = counts_h[ids].astype(np.float64)
    det = sums_h[:, ids].astype(np.float64) / cnt                           # exact integer sums / count = np.mean
    sizes = float(anisotropy) * counts_h[ids].astype(np.float64)             # :507, :516
    return np.ascontiguousarray(det), sizes, ids


def ransac_error_from_sizes(moving_nucleus_size, fixed_nucleus_size):
    """_dock_widget.py:613-618: 16 without sizes (CSV input), else half the sum of the cube roots of the mean sizes."""
    if len(moving_nucleus_size) == 0 or len(fixed_nucleus_size) == 0:
        return 16
    return 0.5 * (np.average(moving_nucleus_size) ** (1 / 3) + np.average(fixed_nucleus_size) ** (1 / 3))
