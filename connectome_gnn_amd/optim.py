"""torch.optim.Adam whose update is ONE HIP launch (csrc/adam.hip, ``cgnn_adam_step``).

Same algorithm, hyper-parameters, ``param_groups`` and ``state_dict`` layout (``step``, ``exp_avg``,
``exp_avg_sq`` per parameter) as ``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` -- what the
reference's demo trains with (demo.py:105-134) -- so it drops into ``Trainer(model, optimizer)``
and checkpoints interchange.  torch's fused multi-tensor Adam is two launches per step and takes
7-43 us for these models' 15 k - 400 k parameters; this is one launch of 5-8 us, and it is
graph-capturable (the step counter lives on the device and is advanced by the same launch).

Restrictions (checked): fp32 CUDA parameters with dense contiguous gradients, no amsgrad /
maximize / differentiable.  All parameters of the optimizer step together and share one step
counter (``state[p]['step']`` is the same 0-dim fp32 device tensor for every parameter).
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class Adam(torch.optim.Adam):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=True)
        self._shared_step = None
        self._arrivals = None
        self._plan = None                 # (key, [(jobs struct, group index), ...], keepalive)

    # ---- state
    def _ensure_state(self):
        dev = None
        for grp in self.param_groups:
            if grp.get("amsgrad") or grp.get("maximize") or grp.get("differentiable"):
                raise ValueError("connectome_gnn_amd.Adam: amsgrad / maximize / differentiable are not supported")
            for p in grp["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise TypeError("connectome_gnn_amd.Adam needs contiguous float32 CUDA parameters")
                dev = p.device
        if dev is None:
            return None
        if self._shared_step is None:
            # adopt a loaded state_dict's counter if there is one (all parameters step together)
            init = 0.0
            for st in self.state.values():
                if "step" in st:
                    init = float(st["step"])
                    break
            self._shared_step = torch.full((), init, dtype=torch.float32, device=dev)
            self._arrivals = torch.zeros(1, dtype=torch.int32, device=dev)
        for grp in self.param_groups:
            for p in grp["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = self._shared_step
        return dev

    def state_dict(self):
        """torch's layout with a PRIVATE 0-dim ``step`` per parameter: the shared device counter is
        an internal of this class, and exporting it aliased would make a torch.optim.Adam that
        loads the checkpoint bump one counter once per parameter each step."""
        sd = super().state_dict()
        sd["state"] = {k: {n: (v.clone() if n == "step" and torch.is_tensor(v) else v) for n, v in st.items()}
                       for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._shared_step = None          # re-adopt the loaded counter at the next step
        self._plan = None

    def _build_plan(self):
        key, launches, keep = [], [], []
        for gi, grp in enumerate(self.param_groups):
            ps = [p for p in grp["params"] if p.grad is not None]
            for lo in range(0, len(ps), _lib.ADAM_MAX_JOBS):
                chunk = ps[lo:lo + _lib.ADAM_MAX_JOBS]
                jb = _lib.CgnnAdamJobs()
                jb.n = len(chunk)
                for i, p in enumerate(chunk):
                    g = p.grad
                    if g.dtype != torch.float32 or not g.is_contiguous() or g.is_sparse:
                        raise TypeError("connectome_gnn_amd.Adam needs dense contiguous float32 gradients")
                    st = self.state[p]
                    jb.numel[i] = p.numel()
                    jb.param[i], jb.grad[i] = p.data_ptr(), g.data_ptr()
                    jb.exp_avg[i], jb.exp_avg_sq[i] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    key.append((p.data_ptr(), g.data_ptr()))
                launches.append((jb, gi))
        return tuple(key), launches

    def _current_key(self):
        return tuple((p.data_ptr(), p.grad.data_ptr()) for grp in self.param_groups for p in grp["params"]
                     if p.grad is not None)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        dev = self._ensure_state()
        if dev is None:
            return loss
        if self._plan is None or self._plan[0] != self._current_key():
            self._plan = self._build_plan()
        lib = _lib.load()
        launches = self._plan[1]
        with _lib.device_guard(dev):
            sp = _lib.stream_ptr(dev)
            for i, (jb, gi) in enumerate(launches):
                grp = self.param_groups[gi]
                b1, b2 = grp["betas"]
                _lib.check(lib.cgnn_adam_step(
                    ctypes.byref(jb), _lib.ptr(self._shared_step), _lib.ptr(self._arrivals),
                    int(i == len(launches) - 1), float(grp["lr"]), float(b1), float(b2), float(grp["eps"]),
                    float(grp["weight_decay"]), sp), "cgnn_adam_step")
        return loss
