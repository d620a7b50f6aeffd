"""`SGD` -- the optimizer of the reference's training notebooks, `keras.optimizers.SGD(lr=0.001, momentum=0.9, decay=0.0,
nesterov=False)` (ssd300_training.ipynb:169), as a torch optimizer whose step is ONE libssdhip launch over every parameter
(csrc/ssdhip_optim.hip, sgd_momentum_kernel) instead of the framework's dozen multi-tensor launches (0.34 ms of an 11 ms SSD300 step,
profiles/r05h_train_step_timeline.json).

Update rule: torch.optim.SGD's (`buf = momentum * buf + grad`, `p -= lr * buf`), which is Keras's (`v = momentum * v - lr * g`,
`p += v`) for a constant learning rate; `weight_decay` adds `weight_decay * p` to the gradient first (Keras expresses the same thing as
`kernel_regularizer=l2(5e-4)` on the loss, models/keras_ssd300.py:274: gradient `2 * 5e-4 * W`).  float32 parameters on a GPU with
gradients of their own memory layout take the fused launch; anything else (CPU tensors, other dtypes, momentum 0) the plain tensor
expressions below."""
from __future__ import annotations

import torch

from . import _native as nat


def _bump_versions(tensors):
    """What an in-place tensor op does to `_version`, for parameters a libssdhip kernel has just updated behind autograd's back."""
    try:
        torch._C._autograd._unsafe_set_version_counter(tuple(tensors), tuple(t._version + 1 for t in tensors))
    except (AttributeError, TypeError):                      # an older framework: a multi-tensor no-op bumps them too
        torch._foreach_add_(list(tensors), 0)


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=0.01, momentum=0.0, weight_decay=0.0, decay=0.0, nesterov=False):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("lr, momentum and weight_decay must be non-negative")
        if decay != 0.0 or nesterov:
            raise ValueError("the reference trains with decay=0.0, nesterov=False (ssd300_training.ipynb:169); neither is implemented")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._tables = {}

    # The device tables hold RAW pointers (parameter, gradient, momentum buffer): everything that can replace one of those tensors
    # drops them, and the per-step key below covers all three addresses as well (the key used to hold the gradients'
    # addresses only -- `load_state_dict` after a step replaced the momentum buffers, the kernel kept updating the freed ones).
    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tables = {}                                     # never pickled pointers: rebuilt on the next step

    def __getstate__(self):
        state = dict(super().__getstate__()) if hasattr(super(), "__getstate__") else dict(self.__dict__)
        state.pop("_tables", None)
        return state

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._tables = {}

    @staticmethod
    def _dense(t):
        return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))

    @staticmethod
    def _same_order(g, p):
        """The two tensors walk memory in the same order: equal strides on every dimension that has more than one entry (a 1 x 1 filter
        is the same memory contiguous or channels_last; only the stride METADATA of its size-1 dimensions differs)."""
        return tuple(g.shape) == tuple(p.shape) and all(a == b for a, b, n in zip(g.stride(), p.stride(), p.shape) if n != 1)

    def _plan(self, gi, group):
        """Which parameters of a group take the fused launch, and its device tables: rebuilt only when a gradient tensor moved (the
        tables hold raw pointers).  Everything per-parameter that can be decided once is decided here: the step itself is host-bound
        (~280 launches in 11 ms), every microsecond of Python in it shows."""
        mom = group["momentum"]
        fused, rest = {}, []
        for p in group["params"]:
            g = p.grad
            if g is None:
                continue
            ok = (mom != 0.0 and p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse
                  and self._dense(p) and self._same_order(g, p) and p.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0)
            if ok:
                fused.setdefault(p.device, []).append(p)
            else:
                rest.append(p)
        tables = []
        flat = lambda t: t.detach().as_strided((t.numel(),), (1,))
        for dev, ps in fused.items():
            bufs = []
            for p in ps:
                st = self.state[p]
                if "momentum_buffer" not in st:               # zeros: the first update is then buf = grad, as torch's clone(grad)
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                bufs.append(st["momentum_buffer"])
            tables.append((nat.sgd_table([flat(p) for p in ps], [flat(p.grad) for p in ps], [flat(b) for b in bufs], dev), tuple(ps)))
        return tables, rest

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            lr, mom, wd = group["lr"], group["momentum"], group["weight_decay"]
            # (the key: which parameters have a gradient, and where the three tensors the kernel touches live)
            state = self.state
            key = []
            for p in group["params"]:
                g = p.grad
                if g is None:
                    key.append(0)
                    continue
                buf = state[p].get("momentum_buffer") if p in state else None
                key.append((g.data_ptr(), p.data_ptr(), 0 if buf is None else buf.data_ptr()))
            key = tuple(key)
            tabs = self.__dict__.setdefault("_tables", {})
            hit = tabs.get(gi)
            if hit is None or hit[0] != key or hit[1] != mom:
                plan = self._plan(gi, group)               # (creates the missing momentum buffers: the key is taken again with them)
                key = tuple(0 if p.grad is None else (p.grad.data_ptr(), p.data_ptr(),
                                                      state[p]["momentum_buffer"].data_ptr() if "momentum_buffer" in state[p] else 0)
                            for p in group["params"])
                hit = (key, mom) + plan
                tabs[gi] = hit
            tables, rest = hit[2], hit[3]
            for table, ps in tables:
                nat.sgd_momentum_step(table, lr, mom, wd)
                _bump_versions(ps)                                  # caches keyed on `_version` (the bf16 shadows) must see the update
            for p in rest:
                g = p.grad
                if wd != 0.0:
                    g = g.add(p, alpha=wd)
                if mom != 0.0:
                    st = self.state[p]
                    buf = st.get("momentum_buffer")
                    if buf is None:
                        buf = st["momentum_buffer"] = torch.clone(g).detach()
                    else:
                        buf.mul_(mom).add_(g)
                    g = buf
                p.add_(g, alpha=-lr)
        return loss
