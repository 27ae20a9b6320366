"""The optimizer step of the training entry (lib/fast_rcnn/train_mv.py:138-146: tf.train.AdamOptimizer(lr)) on this library's kernel:
`Adam` is torch.optim.Adam -- same constructor, same state (`step`, `exp_avg`, `exp_avg_sq` per parameter, so state_dict / load_state_dict and
the `.optim.pt` files of SolverWrapper are interchangeable) -- whose `step()` updates every f32 device parameter that has a gradient with
ONE launch of mv3d_adam_step (csrc/adam.hip) instead of torch's multi-tensor kernels (2.07 ms -> the HBM time of 28 B per element for the
214 M parameters of the 3-view graph).  Anything the kernel does not cover (amsgrad, weight decay, maximize, capturable, non-f32 or host
parameters, sparse gradients) goes to torch's own step."""
import ctypes as C

import numpy as np
import torch

from ._lib import AdamTensor, check, lib


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, **kw):
        kw.pop("fused", None)                                        # (this class IS the fused step)
        super().__init__(params, lr=lr, betas=betas, eps=eps, **kw)
        self.lowp = {}                                                # id(param) -> (16-bit tensor of its shape, stamp callback): see register_lowp

    def register_lowp(self, param, dst, on_written=None):
        """`dst` (float16 / bfloat16, param's number of elements, contiguous) receives the updated `param` in every step of this
        optimizer, in the launch that updates it (the copy a cast at the top of the next step would make).  on_written(param) is called
        after each such step (the owner records that the copy is current)."""
        if dst.numel() != param.numel() or not dst.is_contiguous() or dst.dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("register_lowp: a contiguous float16 / bfloat16 tensor with the parameter's number of elements")
        self.lowp[id(param)] = (dst, on_written)
        self.__dict__.pop("_cache", None)

    def _plain(self, g):
        return not (g.get("amsgrad") or g.get("weight_decay") or g.get("maximize") or g.get("capturable") or g.get("differentiable"))

    def _tables_for(self, key, items, dev):
        """device tables of one launch: the chunk tables depend on the tensors' sizes only (built once per parameter list); the tensor
        table (pointers) is refilled in a pinned staging buffer and copied asynchronously on the step's stream whenever a pointer
        moved -- gradients are fresh allocations in a single-process step (zero_grad drops them), usually at the same addresses"""
        cache = self.__dict__.setdefault("_cache", {})
        ent = cache.get(key)
        if ent is None:
            chunk = lib().mv3d_adam_chunk_elements()
            ct, cf = [], []
            for k, (p, m, v) in enumerate(items):
                n = (p.numel() + chunk - 1) // chunk
                ct.append(np.full(n, k, np.int32)); cf.append(np.arange(n, dtype=np.int32))
            ct_dev = torch.from_numpy(np.concatenate(ct)).to(dev)
            cf_dev = torch.from_numpy(np.concatenate(cf)).to(dev)
            nbytes = C.sizeof(AdamTensor) * len(items)
            ent = {"ct": ct_dev, "cf": cf_dev, "n": int(ct_dev.numel()), "pin": torch.empty(nbytes, dtype=torch.uint8).pin_memory(),
                   "dev": torch.empty(nbytes, dtype=torch.uint8, device=dev), "ptrs": None}
            cache[key] = ent
        lp = [self.lowp.get(id(p)) for p, _, _ in items]
        ptrs = tuple((p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr(), l[0].data_ptr() if l else 0) for (p, m, v), l in zip(items, lp))
        kinds = {l[0].dtype for l in lp if l}
        if len(kinds) > 1:
            raise ValueError("register_lowp: one 16-bit type per optimizer")
        ent["lowp"] = 0 if not kinds else (1 if kinds.pop() == torch.float16 else 2)
        if ptrs != ent["ptrs"]:
            arr = (AdamTensor * len(items)).from_buffer(ent["pin"].numpy())
            for k, ((p, _, _), q) in enumerate(zip(items, ptrs)):
                arr[k] = AdamTensor(q[0], q[1], q[2], q[3], p.numel(), q[4] or None)
            ent["dev"].copy_(ent["pin"], non_blocking=True)           # (stream-ordered before the launch below)
            ent["ptrs"] = ptrs
        return ent

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        rest = []
        for gi, g in enumerate(self.param_groups):
            items = []
            for p in g["params"]:
                if p.grad is None:
                    continue
                ok = self._plain(g) and p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and not p.grad.is_sparse \
                    and p.is_contiguous() and p.grad.is_contiguous()
                if not ok:
                    rest.append(gi)
                    items = None
                    break
                st = self.state[p]
                if len(st) == 0:                                      # torch.optim.Adam's own lazy state
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                items.append((p, st["exp_avg"], st["exp_avg_sq"]))
            if not items:
                continue
            # one step count per launch: parameters that joined later (a gradient for the first time) keep their own count in `state`;
            # the launch is cut by count so that every tensor gets its own bias corrections
            by_step = {}
            for it in items:
                by_step.setdefault(int(float(self.state[it[0]]["step"])), []).append(it)
            for t0, its in by_step.items():
                dev = its[0][0].device
                ent = self._tables_for((gi, tuple(id(p) for p, _, _ in its)), its, dev)
                t_dev, ct_dev, cf_dev, n = ent["dev"], ent["ct"], ent["cf"], ent["n"]
                b1, b2 = g["betas"]
                check(lib().mv3d_adam_step(C.c_void_p(t_dev.data_ptr()), C.c_void_p(ct_dev.data_ptr()), C.c_void_p(cf_dev.data_ptr()), n,
                                           float(g["lr"]), float(b1), float(b2), float(g["eps"]), t0 + 1, ent["lowp"],
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mv3d_adam_step")
                for p, _, _ in its:
                    self.state[p]["step"] += 1
                    l = self.lowp.get(id(p))
                    if l and l[1]:
                        l[1](p)
        if rest:                                                      # whatever the kernel does not cover: torch's own update for those groups
            keep = self.param_groups
            try:
                self.param_groups = [keep[i] for i in sorted(set(rest))]
                super().step()
            finally:
                self.param_groups = keep
        return loss
