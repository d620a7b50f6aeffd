"""Drop-in for the reference's `ssd_encoder_decoder/matching_utils.py`, computed on the GPU
(`ssdhip_match_bipartite_greedy`, `ssdhip_match_multi`; include/ssdhip.h).

`weight_matrix` may be a NumPy array (result: NumPy int64 arrays, as `np.argmax` returns) or a CUDA torch tensor
(result: CUDA int64 tensors).  There is no CPU path.
"""
from __future__ import annotations


from .. import _native as nat


def _prep(weight_matrix):
    import torch
    as_np = not torch.is_tensor(weight_matrix)
    w = nat._float_device(weight_matrix, 'weight_matrix').to(torch.float64)
    if w.dim() != 2:
        raise ValueError("weight_matrix must be a 2D array of shape (m, n)")
    return w.contiguous(), as_np


def match_bipartite_greedy(weight_matrix):
    '''Reference :22-79: greedy bipartite matching of the `m` rows (ground truth boxes) to the `n` columns (anchors),
    `m <= n`: `m` times take the largest remaining entry, record its column for its row, zero that row and column.
    Returns the matched column of every row, shape `(m,)`.  The input is not modified.'''
    import torch
    w, as_np = _prep(weight_matrix)
    out = nat.match_bipartite_greedy(w).to(torch.int64)
    return out.cpu().numpy() if as_np else out


def match_multi(weight_matrix, threshold):
    '''Reference :81-116: every column is matched to its (first) arg-max row if that entry is `>= threshold`.
    Returns `(row_indices, column_indices)` of the matches, columns ascending.'''
    import torch
    w, as_np = _prep(weight_matrix)
    if w.shape[0] == 0:
        raise ValueError("attempt to get argmax of an empty sequence")
    gt, col = nat.match_multi(w, threshold)
    gt, col = gt.to(torch.int64), col.to(torch.int64)
    return (gt.cpu().numpy(), col.cpu().numpy()) if as_np else (gt, col)
