"""Optimizer step of the training path: `FlatAdamW` keeps parameters, gradients and
both AdamW moments in ONE flat fp32 buffer each, so that

  * the gradient all-reduce runs on contiguous slices of the flat gradient (no pack / unpack copies,
    `dp.GradReducer` launches each slice as soon as backward has filled it), and
  * AdamW (valle/models/valle_ar.py:182-194), the global-norm clip (`gradient_clip_val`,
    valle/train_model.py:31-32) and the 1/world mean of the all-reduce are one elementwise HIP pass
    (`vh_adamw_flat`: one partial-norm launch + one update launch, no host read of the norm).

Parameters that receive no gradient in a step are skipped and keep their own step count, as torch.optim.AdamW
does for `grad is None` (NAR trains one stage per step, valle_nar.py:76).  "Received a gradient" means: autograd
accumulated a gradient for the parameter (post-accumulate hook) or a gradient tensor was assigned to `p.grad`;
writing into the flat view by hand without either is not seen.  After zero_grad() — or step(zero_grad=True) — the
gradients are `None`, as with torch's set_to_none: the backward functions of this package then write each
parameter's first gradient of the step directly into its slice of the flat buffer (`grad_out`), which autograd
adopts as `p.grad` without an accumulation launch; later gradients of the same step (gradient accumulation) are
added onto it in place by autograd.

It subclasses `torch.optim.Optimizer` only for the `param_groups` contract the reference's
`CosineAnnealingWarmRestarts` scheduler drives (`lr` is read from the group at every step).
"""
from __future__ import annotations

import weakref

import torch

from . import _lib
from ._lib import ptr

# parameter storage address -> (weak FlatAdamW, slot index): lets backward write a parameter's FIRST gradient of a step
# straight into its slice of the flat gradient (grad_out), so that autograd's AccumulateGrad adopts that tensor — no
# `grad += new` launch per parameter (≈150 per step of the 12-layer models), no zeros fill for the accumulate-type
# kernels (bias column sums, embedding scatter-adds, LayerNorm weight gradients)
GRAD_SLOTS = {}


def grad_out(param, like=None, zero=False):
    """Output tensor for `param`'s gradient in a backward function: its slice of a FlatAdamW's flat gradient when this
    is the parameter's first gradient since the buffer was zeroed (`param.grad is None`), else a fresh tensor.
    zero=True: the kernel ACCUMULATES into it (the slice is zero after zero_grad(); a fresh tensor is zero-filled)."""
    ref = param if param is not None else like
    ent = GRAD_SLOTS.get(param.data_ptr()) if param is not None else None
    if ent is not None and param.grad is None:
        opt = ent[0]()
        if opt is not None and opt._claim(ent[1]):
            view = opt.grad_view(opt.slots[ent[1]])
            if zero and not opt._clean[ent[1]]:
                view.zero_()
            opt._clean[ent[1]] = False
            return view
    return (torch.zeros_like if zero else torch.empty_like)(ref, memory_format=torch.contiguous_format)

ALIGN = 64  # floats: every parameter slot starts on a 256-byte boundary (the GEMM kernels need 16 bytes; the
            # optimizer kernel maps 64-float blocks to slots for per-parameter step counts)


def flat_layout(params):
    """[(param, offset, numel)] in REVERSE registration order (backward produces gradients roughly
    last layer first, so the first slices of the flat buffer fill first) and the padded total."""
    slots, off = [], 0
    for p in reversed(list(params)):
        slots.append((p, off, p.numel()))
        off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
    return slots, off


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError('FlatAdamW: no trainable parameters')
        dev = params[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in params):
            raise _lib.VhError('FlatAdamW: parameters must be fp32 on one device')
        if dev.type != 'cuda':
            raise _lib.VhError('FlatAdamW: move the model to its HIP device first (model.to("cuda")); the '
                               'optimizer re-homes the parameters into one flat device buffer')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.slots, self.numel = flat_layout(params)
        f32 = dict(device=dev, dtype=torch.float32)
        self.flat_param = torch.zeros(self.numel, **f32)
        self.flat_grad = torch.zeros(self.numel, **f32)
        self.exp_avg = torch.zeros(self.numel, **f32)
        self.exp_avg_sq = torch.zeros(self.numel, **f32)
        self.grad_norm = torch.zeros(1, **f32)
        self._ws = None
        self.steps = 0
        # torch.optim.AdamW semantics for parameters that receive no gradient in a step (`grad is None`:
        # skipped, and `step` is counted per parameter): post-accumulate hooks mark the slots autograd touched
        self.slot_steps = [0] * len(self.slots)
        self._touched = [False] * len(self.slots)
        self._block_slot = None
        self._slot_step_dev = torch.zeros(len(self.slots), device=dev, dtype=torch.int32)
        # parameters that already live in another FlatAdamW's flat buffer (configure_optimizers called twice, a
        # resume): detach the old optimizer first — its hooks and registry entries would otherwise stay live
        for p in params:
            ent = GRAD_SLOTS.get(p.data_ptr())
            old = ent[0]() if ent is not None else None
            if old is not None and old is not self:
                old.close()
        me = weakref.ref(self)                 # the hooks must not keep the optimizer (and its buffers) alive

        def touch(_p, i):
            opt = me()
            if opt is not None:
                opt._touched[i] = True
        self._hooks = [p.register_post_accumulate_grad_hook(lambda _p, i=i: touch(_p, i))
                       for i, (p, _, _) in enumerate(self.slots)]
        with torch.no_grad():
            for p, off, n in self.slots:
                view = self.flat_param[off:off + n].view_as(p)
                view.copy_(p)
                p.data = view                      # the module now computes on the flat buffer
        for i, (p, _, _) in enumerate(self.slots):
            GRAD_SLOTS[p.data_ptr()] = (weakref.ref(self), i)
        # asynchronous copies of the device error flag, one per step still unverified: (event, pinned copy, host
        # counters before that step) — see check_errors
        self._flag_log = []
        self._flag_pool = []
        self._claimed = [False] * len(self.slots)  # slice handed to backward as an output this step (grad_out)
        self._clean = [True] * len(self.slots)     # slice known to be all zero
        self._release_grads()

    def check_errors(self, wait=True):
        """Raise the IndexError of a step whose kernels saw an out-of-range id.  That step's update was not applied on
        the device (the update launch checks the flag itself), nor any later one — the flag is sticky until it is read
        here — so the host's step counters are rolled back to what they were before the flagged step, and the gradient
        bookkeeping is reset (the guarded launch cleared the flat gradient if asked to; it is cleared here otherwise):
        a caller that catches the error and skips the batch continues from a consistent state.
        wait=False: only look at flag copies that have already arrived — the copy of the previous step has, by the
        time the next step's backward is over; wait=True (end of training, tests): synchronise."""
        while self._flag_log:
            event, host, snap = self._flag_log[0]
            if not (wait or event.query()):
                return
            event.synchronize()
            code = int(host[0])
            self._flag_log.pop(0)
            self._flag_pool.append(host)
            if code:
                self.steps, self.slot_steps = snap
                for _, h, _ in self._flag_log:             # later steps were skipped on the device as well
                    self._flag_pool.append(h)
                self._flag_log = []
                self.flat_grad.zero_()
                self._clean = [True] * len(self.slots)
                self._touched = [False] * len(self.slots)
                self._release_grads()
                _lib.raise_device_errors(self.flat_param.device, code=code)

    def close(self):
        """Detach from the parameters: remove the autograd hooks (their closures keep this optimizer — four flat
        buffers of the model's size — alive as long as the parameters live) and this optimizer's entries of the
        gradient-slot registry.  Called automatically when another FlatAdamW is built over the same parameters."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p, _, _ in self.slots:
            ent = GRAD_SLOTS.get(p.data_ptr())
            if ent is not None and ent[0]() in (self, None):
                del GRAD_SLOTS[p.data_ptr()]

    def __del__(self):
        # drop this optimizer's entries from the address registry (an address can be reused by another tensor)
        try:
            self.close()
        except Exception:
            pass

    def _block_map(self):
        if self._block_slot is None:
            m = torch.empty(self.numel // ALIGN, dtype=torch.int32)
            for i, (_, off, n) in enumerate(self.slots):
                m[off // ALIGN:(off + n + ALIGN - 1) // ALIGN] = i
            self._block_slot = m.to(self.flat_param.device)
        return self._block_slot

    def sync_touched(self):
        """Data parallel: a slot is updated iff SOME rank produced a gradient for it (its all-reduced gradient is
        then the same everywhere), so every rank takes the same decision and the replicas stay identical.

        The device error flag travels with the flags, and the decision it carries is taken HERE, collectively: the MAX
        over ranks is read on the host in this call (the flags are read anyway), so when any rank's kernels saw a bad id
        during this step's forward / backward — the range-checking kernels all run before this point — EVERY rank drops
        the step in this very call: gradients cleared, flag cleared, host counters untouched (step() has not run), and
        the IndexError raised on all ranks together.  No rank can run ahead into the next bucket's collective while a
        peer raises, and no stale flag is left to be re-raised from a peer at the next step."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        flag = _lib.err_flag(self.flat_param.device)
        t = torch.cat([torch.tensor(self._touched, dtype=torch.int32, device=self.flat_param.device), flag])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        vals = t.tolist()                                  # (host synchronisation: once per optimizer step, world > 1 only)
        self._touched = [bool(v) for v in vals[:-1]]
        code = int(vals[-1])
        if code:
            flag.zero_()
            self.flat_grad.zero_()
            self._clean = [True] * len(self.slots)
            self._touched = [False] * len(self.slots)
            self._release_grads()
            _lib.raise_device_errors(self.flat_param.device, code=code)

    # ---- gradient views -------------------------------------------------------------------------
    def grad_view(self, slot):
        p, off, n = slot
        return self.flat_grad[off:off + n].view_as(p)

    def _claim(self, i):
        if self._claimed[i]:
            return False
        self._claimed[i] = True
        return True

    def _release_grads(self):
        """After the flat gradient has been zeroed: `p.grad = None` (torch's set_to_none semantics), so that the next
        backward's first gradient of each parameter is written straight into its slice (grad_out)."""
        for slot in self.slots:
            slot[0].grad = None
        self._claimed = [False] * len(self.slots)

    def gather_grads(self):
        """Every gradient into its slice of the flat buffer: autograd either adopted the slice (grad_out) or
        accumulated into it in place; a gradient tensor that lives elsewhere (assigned from outside, or cloned by
        autograd) is copied in; a parameter without a gradient contributes zeros."""
        for i, slot in enumerate(self.slots):
            p, view = slot[0], self.grad_view(slot)
            if p.grad is None:
                if not self._clean[i]:
                    view.zero_()
                    self._clean[i] = True
                continue
            if p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                self._touched[i] = True            # a gradient assigned from outside counts as present
                p.grad = view
            self._clean[i] = False

    def zero_grad(self, set_to_none: bool = True):
        self.flat_grad.zero_()
        self._clean = [True] * len(self.slots)
        self._release_grads()

    # ---- the step -------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None, *, grad_scale: float = 1.0, max_norm: float = 0.0, zero_grad: bool = False):
        """One AdamW update of every parameter.  grad_scale multiplies the gradients first (1/world
        after a sum all-reduce), max_norm > 0 clips their global norm as clip_grad_norm_ does.
        Returns the (device) gradient norm after scaling, before clipping."""
        if closure is not None:
            raise _lib.VhError('FlatAdamW.step: closures are not supported')
        self.gather_grads()
        # ids / targets already resident on the device are range-checked by the kernels: a bad one must not reach the
        # parameters.  The update launch checks the flag itself (`guard`) and leaves everything untouched when it is
        # set; the host reads the flag through an asynchronous copy and raises at the NEXT step (or check_errors()),
        # so there is no host synchronisation per step and the next forward is enqueued while this step still runs.
        self.check_errors(wait=False)
        flag = _lib.err_flag(self.flat_param.device)
        g = self.param_groups[0]
        lib = _lib.lib()
        if self._ws is None:
            self._ws = torch.empty(lib.vh_adamw_ws_bytes() // 8, device=self.flat_param.device, dtype=torch.float64)
        snap = (self.steps, list(self.slot_steps))  # restored by check_errors if this step turns out to be guarded off
        self.steps += 1
        # per-slot step counts: a slot without a gradient this step is skipped (no decay, no moments) and its
        # count stands still; when every slot has always been updated the kernel takes the uniform fast path
        now = [(c + 1) if t else 0 for c, t in zip(self.slot_steps, self._touched)]
        self.slot_steps = [max(c, n) for c, n in zip(self.slot_steps, now)]
        uniform = all(n == self.steps for n in now)
        block_slot = slot_step = None
        if not uniform:
            block_slot = self._block_map()
            self._slot_step_dev.copy_(torch.tensor(now, dtype=torch.int32))
            slot_step = self._slot_step_dev
        self._touched = [False] * len(self.slots)
        _lib.check(lib.vh_adamw_flat(
            ptr(self.flat_param), ptr(self.flat_grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.numel,
            float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
            float(g['weight_decay']), self.steps, float(grad_scale), float(max_norm), int(zero_grad),
            ptr(self._ws), ptr(self.grad_norm), ptr(block_slot), ptr(slot_step), ptr(flag),
            torch.cuda.current_stream().cuda_stream), 'vh_adamw_flat')
        host = self._flag_pool.pop() if self._flag_pool else torch.zeros(1, dtype=torch.int32).pin_memory()
        host.copy_(flag, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        self._flag_log.append((event, host, snap))
        if zero_grad:                              # the kernel cleared the flat gradient
            self._clean = [True] * len(self.slots)
            self._release_grads()
        from . import engine
        engine.bump_weights_epoch()                # the update bypasses torch's version counters
        return self.grad_norm

    # ---- checkpointing: the four flat buffers + the step count ------------------------------------
    def state_dict(self):
        return {'steps': self.steps, 'slot_steps': list(self.slot_steps), 'exp_avg': self.exp_avg,
                'exp_avg_sq': self.exp_avg_sq,
                'param_groups': [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.steps = int(sd['steps'])
        self.slot_steps = list(sd.get('slot_steps', [self.steps] * len(self.slots)))
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        for g, saved in zip(self.param_groups, sd['param_groups']):
            g.update(saved)
