"""Drop-in for the reference's `bounding_box_utils/bounding_box_utils.py`, computed on the GPU.

Same function names, arguments, error messages' conditions and result dtypes as the reference
(`convert_coordinates` :24-87, `convert_coordinates2` :89-117, `intersection_area` :119-224,
`intersection_area_` :226-280, `iou` :283-383); the arithmetic runs in libssdhip.so
(`ssdhip_convert_coordinates`, `ssdhip_box_overlap`; include/ssdhip.h) with NumPy's dtype rules.

Inputs may be NumPy arrays (copied to the current GPU; the result comes back as a NumPy array) or CUDA
torch tensors (used in place; the result stays on the GPU).  There is no CPU path.
"""
from __future__ import annotations


from .. import _native as nat

_CONVERSIONS = ('minmax2centroids', 'centroids2minmax', 'corners2centroids', 'centroids2corners', 'minmax2corners',
                'corners2minmax')


def _is_np(a):
    import torch
    return not torch.is_tensor(a)


def _back(t, as_numpy):
    return t.cpu().numpy() if as_numpy else t


def convert_coordinates(tensor, start_index, conversion, border_pixels='half'):
    '''Reference :24-87.  Returns a float64 copy of `tensor` whose four coordinates starting at `start_index` of
    the last axis are converted; the right-hand sides are evaluated in the input's dtype, as NumPy does.'''
    if conversion not in _CONVERSIONS:
        raise ValueError("Unexpected conversion value. Supported values are 'minmax2centroids', 'centroids2minmax', "
                         "'corners2centroids', 'centroids2corners', 'minmax2corners', and 'corners2minmax'.")
    if border_pixels not in nat.BORDER:
        raise ValueError("`border_pixels` must be one of 'half', 'include' and 'exclude'")
    as_np = _is_np(tensor)
    t = nat._float_device(tensor, 'tensor')
    ind = int(start_index)
    if ind < 0:
        ind += t.shape[-1]
    if t.dim() < 1 or ind < 0 or ind + 4 > t.shape[-1]:
        raise IndexError("the last axis must hold four coordinates starting at `start_index`")
    return _back(nat.convert_coordinates(t, ind, conversion, border_pixels), as_np)


def convert_coordinates2(tensor, start_index, conversion):
    '''Reference :89-117: the matrix-product formulation, 'centroids' <-> 'minmax' only.  It multiplies the float64 copy
    by a constant matrix of 0 / +-0.5 / +-1 entries, which rounds exactly like `convert_coordinates` on float64 input.'''
    if conversion not in ('minmax2centroids', 'centroids2minmax'):
        raise ValueError("Unexpected conversion value. Supported values are 'minmax2centroids' and 'centroids2minmax'.")
    import torch
    as_np = _is_np(tensor)
    t = nat._float_device(tensor, 'tensor').to(torch.float64)
    return _back(nat.convert_coordinates(t, int(start_index), conversion, 'half'), as_np)


def _prepare(boxes1, boxes2, coords, mode):
    b1 = nat._float_device(boxes1, 'boxes1')
    b2 = nat._float_device(boxes2, 'boxes2')
    if b1.dim() > 2:
        raise ValueError("boxes1 must have rank either 1 or 2, but has rank {}.".format(b1.dim()))
    if b2.dim() > 2:
        raise ValueError("boxes2 must have rank either 1 or 2, but has rank {}.".format(b2.dim()))
    if b1.dim() == 1:
        b1 = b1.unsqueeze(0)
    if b2.dim() == 1:
        b2 = b2.unsqueeze(0)
    if not (b1.shape[1] == b2.shape[1] == 4):
        raise ValueError("All boxes must consist of 4 coordinates, but the boxes in `boxes1` and `boxes2` have {} and {} "
                         "coordinates, respectively.".format(b1.shape[1], b2.shape[1]))
    if mode not in ('outer_product', 'element-wise'):
        raise ValueError("`mode` must be one of 'outer_product' and 'element-wise', but got '{}'.".format(mode))
    if coords not in nat.COORDS:
        raise ValueError("Unexpected value for `coords`. Supported values are 'minmax', 'corners' and 'centroids'.")
    if mode == 'element-wise' and b1.shape[0] != b2.shape[0] and 1 not in (b1.shape[0], b2.shape[0]):
        raise ValueError("operands could not be broadcast together with shapes ({},4) ({},4)".format(b1.shape[0], b2.shape[0]))
    return b1.contiguous(), b2.contiguous()


def intersection_area(boxes1, boxes2, coords='centroids', mode='outer_product', border_pixels='half'):
    '''Reference :119-224: intersection areas `(m, n)` ('outer_product') or `(m,)` ('element-wise'); the side lengths
    see `border_pixels`.'''
    as_np = _is_np(boxes1) and _is_np(boxes2)
    b1, b2 = _prepare(boxes1, boxes2, coords, mode)
    return _back(nat.box_overlap(1, b1, b2, coords, mode, border_pixels), as_np)


def intersection_area_(boxes1, boxes2, coords='corners', mode='outer_product', border_pixels='half'):
    '''Reference :226-280: `intersection_area` without argument checks and without 'centroids' support.'''
    if coords not in ('corners', 'minmax'):
        raise ValueError("intersection_area_ supports 'corners' and 'minmax' only")
    return intersection_area(boxes1, boxes2, coords=coords, mode=mode, border_pixels=border_pixels)


def iou(boxes1, boxes2, coords='centroids', mode='outer_product', border_pixels='half'):
    '''Reference :283-383: intersection over union, `(m, n)` or `(m,)`.  As in the reference (:345) the intersection is
    always taken with the 'half' convention; only the two box areas see `border_pixels`.'''
    as_np = _is_np(boxes1) and _is_np(boxes2)
    b1, b2 = _prepare(boxes1, boxes2, coords, mode)
    return _back(nat.box_overlap(0, b1, b2, coords, mode, border_pixels), as_np)
