"""Device-resident images behind a NumPy-looking handle.

The reference passes NumPy arrays between its stages (disk_list, frame_circularized,
cc, ...).  Here those objects stay in HBM between stages; DeviceImage gives callers the
ndarray surface they expect (shape, dtype, indexing, np.asarray) by copying to the host
lazily and once.
"""
import numpy as np
import torch


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError('no GPU visible: the SHG hot path runs on MI355X only (no CPU fallback)')
    return torch.device('cuda', torch.cuda.current_device())


class DeviceImage:
    __array_priority__ = 100

    def __init__(self, tensor, row_factor=None, minmax=None):
        """row_factor (float64 GPU tensor [h], optional): the image is then the float64 array
        tensor[y, x] * row_factor[y] -- what the reference's removeVignette returns (solex_util.py:654) --
        kept factored so that it never has to be materialised in HBM.
        minmax (int32 GPU tensor [2] = {min, max}, optional): the image's extrema as the extraction kernel gathered them
        (shg_extract_columns_minmax); the warp clips to them without another pass over the image."""
        if not isinstance(tensor, torch.Tensor) or not tensor.is_cuda:
            raise TypeError('DeviceImage wraps a GPU tensor')
        self.t = tensor
        self.row_factor = row_factor
        self.minmax = minmax
        self._host = None

    # ndarray surface -------------------------------------------------------
    @property
    def shape(self):
        return tuple(self.t.shape)

    @property
    def ndim(self):
        return self.t.dim()

    @property
    def dtype(self):
        if self.row_factor is not None:
            return np.dtype(np.float64)
        return np.dtype(str(self.t.dtype).replace('torch.', ''))

    def numpy(self):
        if self._host is None:
            host = self.t.cpu().numpy()
            if self.row_factor is not None:
                host = host * self.row_factor.cpu().numpy().reshape((-1, 1))
            self._host = host
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, idx):
        return self.numpy()[idx]

    def __len__(self):
        return self.t.shape[0]

    def astype(self, dtype):
        return self.numpy().astype(dtype)

    def __truediv__(self, other):
        return self.numpy() / other

    def __eq__(self, other):
        return self.numpy() == np.asarray(other)

    def __repr__(self):
        return 'DeviceImage(%s, %s, %s)' % (self.shape, self.dtype, self.t.device)


def to_device_u16(img, device=None):
    """GPU uint16 tensor (2-D, unit column stride) from a DeviceImage, tensor or ndarray."""
    if isinstance(img, DeviceImage):
        if img.row_factor is not None:
            raise TypeError('this stage needs a uint16 image, got a factored float64 one')
        return img.t
    if isinstance(img, torch.Tensor):
        if not img.is_cuda:
            raise RuntimeError('expected a GPU tensor')
        return img
    arr = np.asarray(img)
    if arr.dtype != np.uint16:
        raise TypeError('expected a uint16 image, got %s' % arr.dtype)
    return torch.from_numpy(np.ascontiguousarray(arr)).to(device or default_device())


def u16_from_unit_float(image):
    """The reference hands `disk / 65536` (float64) to correct_image (Solex_recon.py:123,
    ellipse_to_circle.py:299).  Recover the exact uint16 disk, or refuse."""
    if isinstance(image, (DeviceImage, torch.Tensor)):
        return image
    arr = np.asarray(image)
    if arr.dtype == np.uint16:
        return arr
    scaled = arr * 65536.0
    back = np.rint(scaled)
    if arr.size and (np.abs(scaled - back).max() != 0 or back.min() < 0 or back.max() > 65535):
        raise ValueError('correct_image expects uint16 data or uint16/65536 (16-bit assumption of the reference)')
    return back.astype(np.uint16)
