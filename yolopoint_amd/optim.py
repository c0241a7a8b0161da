"""The reference's optimizer -- torch.optim.Adam(model.parameters(), lr) (src/train.py:88, stepped at :252) -- as ONE launch.

FlatAdam keeps the parameters, their gradients (dp.GradAllReducer's arena: p.grad are views of it) and both moments as four flat fp32
arrays with the same layout and runs yp_adam_step over them: one pass, one launch, instead of torch's multi-tensor kernels over
30-odd tensors each (6 launches, 0.38 ms for YOLOPoint-s).  Same update rule, fp32, bias corrections in double on the host.
It is a torch.optim.Optimizer: param_groups / lr schedulers / the optimizer-step hooks work as with torch.optim.Adam.
"""
import ctypes as C

import torch

from . import _hip
from ._hip import lib, check


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, reducer, params=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, all_params=None):
        """reducer: dp.GradAllReducer over the parameters to optimize (its arena is the gradient array; bind_grads() must be in effect
        when step() runs).  params: the same parameters in the order torch.optim.Adam would have been given them (model.parameters()):
        it numbers the entries of state_dict(); default: the reducer's order.
        all_params: every parameter of the model in model.parameters() order, frozen ones included -- the reference hands all of them to
        Adam (train.py:88), which numbers its state by that position and simply skips parameters without a gradient; with it a checkpoint
        of a partly frozen model (freeze_layers) lines up with torch.optim.Adam's."""
        self.reducer = reducer
        self._numbering = None if all_params is None else {id(p): i for i, p in enumerate(all_params)}
        self._n_all = None if all_params is None else len(self._numbering)
        self.flat_p = reducer.flatten_parameters()
        order = list(params) if params is not None else list(reducer.params)
        if set(id(p) for p in order) != set(id(p) for p in reducer.params):
            raise _hip.YpError("FlatAdam: `params` must be exactly the reducer's parameters")
        super().__init__(order, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.flat_m, self.flat_v = torch.zeros_like(self.flat_p), torch.zeros_like(self.flat_p)
        self.steps = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self.reducer.grads_bound():
            raise _hip.YpError("FlatAdam.step: the gradients are not the reducer's bucket views (call reducer.bind_grads() instead of zero_grad())")
        base = self.flat_p.untyped_storage().data_ptr()
        if any(p.untyped_storage().data_ptr() != base for p in (self.reducer.params[0], self.reducer.params[-1])):
            raise _hip.YpError("FlatAdam.step: the parameters no longer live in this optimizer's arena (another reducer / optimizer re-flattened them)")
        g = self.param_groups[0]
        self.steps += 1
        with torch.cuda.device(self.flat_p.device):
            check(lib().yp_adam_step(self.flat_p.data_ptr(), self.reducer.arena.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(), self.flat_p.numel(),
                                     float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self.steps,
                                     _hip.stream_ptr()))
        return loss

    def state_dict(self):
        """torch.optim.Adam's layout (per-parameter exp_avg / exp_avg_sq / step, parameters numbered in param_groups order), so that a
        checkpoint written here resumes under torch.optim.Adam and vice versa."""
        state, o = {}, {}
        for off, (_, entries) in zip(self.reducer.bucket_offsets, self.reducer.buckets):
            for p, po, n in entries:
                o[id(p)] = (off + po, n)
        for i, p in enumerate(self.param_groups[0]["params"]):
            a, n = o[id(p)]
            state[self._index(i, p)] = {"step": torch.tensor(float(self.steps)), "exp_avg": self.flat_m[a:a + n].view_as(p).clone(),
                                        "exp_avg_sq": self.flat_v[a:a + n].view_as(p).clone()}
        groups = [{k: v for k, v in self.param_groups[0].items() if k != "params"}]
        groups[0]["params"] = list(range(self._n_all if self._n_all is not None else len(self.param_groups[0]["params"])))
        return {"state": state, "param_groups": groups}

    def _index(self, i, p):
        return i if self._numbering is None else self._numbering[id(p)]

    def load_state_dict(self, sd):
        o = {}
        for off, (_, entries) in zip(self.reducer.bucket_offsets, self.reducer.buckets):
            for p, po, n in entries:
                o[id(p)] = (off + po, n)
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = v
        with torch.no_grad():
            for i, p in enumerate(self.param_groups[0]["params"]):
                st = sd["state"].get(self._index(i, p))
                if st is None:
                    continue
                a, n = o[id(p)]
                self.flat_m[a:a + n].view_as(p).copy_(st["exp_avg"])
                self.flat_v[a:a + n].view_as(p).copy_(st["exp_avg_sq"])
                self.steps = int(st["step"])
