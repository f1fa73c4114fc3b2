"""Stale-cache guard for the kernel-ready parameter copies (`layer.LayerWeights`).

The reference reads its `nn.Parameter`s directly on every call (`HermNet/hermnet.py:118-131`), so ANY way of changing a
weight takes effect at once.  Here the fused eval() path keeps derived copies (MFMA operand order, folded LayerNorm affine,
transposes) keyed on each parameter's (identity, version counter, address) -- and a write through `.data` (EMA / SWA swaps,
old-style `p.data.copy_(...)`) changes none of those.  Two mechanisms close the gap:

* `HVNet.invalidate_caches()` runs on `load_state_dict`, `_apply` (`.to()`, `.cuda()`, `.float()` ...) and `train()` /
  `eval()` transitions;
* every eval() forward launches ONE small kernel (`hermnet_param_guard`) that fingerprints all parameters and buffers on
  the device and compares with the fingerprints recorded when the copies were built.  On a difference it overwrites a
  cached bias that every result depends on with NaN -- the step yields NaN, never the old numbers, with no host
  synchronisation -- and raises a flag that the NEXT forward reads (asynchronously copied to pinned memory): that call
  warns, rebuilds the copies and computes with the new weights.
"""
import ctypes
import warnings

import torch

from . import _lib


def _no_guard():
    return None


class ParamGuard(object):
    def __init__(self, module):
        self.module = module
        self.key = None                 # (addresses, sizes) the pointer table was built for
        self.armed_for = None           # the cache stamp the fingerprints were recorded for
        self._event = None
        self._flag_host = None
        self._warned_skipped = False

    # A guard belongs to ONE module instance and holds device state (a HIP event, pinned memory, pointer tables): copies and
    # pickles of the module get none -- `HVNet._refresh_weights` makes a fresh, unarmed one on the copy's first eval() forward.
    # (`copy.deepcopy(model)` is how `torch.optim.swa_utils.AveragedModel` and best-model snapshots take their copy;
    # `torch.save(model)` pickles `module.__dict__`: both met "cannot pickle 'Event' object" before.)
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_no_guard, ())

    def _tensors(self):
        """Every parameter and buffer the fingerprint covers: 32-bit words on the device.  A tensor that is NOT covered (another
        element size, not contiguous, not on the GPU) is named in a warning once -- a write through `.data` to such a tensor
        is not detected and needs `model.invalidate_caches()`."""
        m = self.module
        every = [(n, t) for n, t in list(m.named_parameters()) + list(m.named_buffers()) if t is not None and t.numel() > 0]
        ok = lambda t: t.is_cuda and t.element_size() == 4 and t.is_contiguous()
        skipped = [n for n, t in every if not ok(t)]
        if skipped and not self._warned_skipped:
            self._warned_skipped = True
            warnings.warn("hermnet_amd: the stale-cache guard does not cover %d tensor(s) (not 32-bit, not contiguous or not on "
                          "the GPU): %s -- after writing them through `.data` call model.invalidate_caches()"
                          % (len(skipped), ", ".join(skipped[:6]) + (" ..." if len(skipped) > 6 else "")),
                          RuntimeWarning, stacklevel=4)
        return [t for _, t in every if ok(t)]

    def _table(self, dev):
        ts = self._tensors()
        key = tuple((t.data_ptr(), t.numel()) for t in ts)
        if key != self.key:
            # one workgroup per chunk of at most CHUNK words (hermnet_param_guard)
            CHUNK = 4096
            chunks = [(ptr + 4 * off, min(CHUNK, n - off)) for ptr, n in key for off in range(0, n, CHUNK)]
            self.ptrs = torch.tensor([c[0] for c in chunks], dtype=torch.long, device=dev)
            self.counts = torch.tensor([c[1] for c in chunks], dtype=torch.long, device=dev)
            self.fp = torch.zeros(max(len(chunks), 1), dtype=torch.int32, device=dev)
            self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
            self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self.key, self.armed_for, self._event = key, None, None
        return len(key)

    def _launch(self, check, poison):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.load().hermnet_param_guard(self.ptrs.data_ptr(), self.counts.data_ptr(), int(self.counts.numel()),
                                                   self.fp.data_ptr(), 1 if check else 0,
                                                   None if poison is None else poison.data_ptr(), self.flag.data_ptr(), stream),
                   "hermnet_param_guard")

    def tripped(self, wait=False):
        """Has a check found the parameters changed behind the cached copies?  Non-blocking unless `wait` (False also while
        the last check's flag is still on its way to the host)."""
        if self._event is None:
            return False
        if wait:
            self._event.synchronize()
        elif not self._event.query():
            return False
        return bool(int(self._flag_host[0]))

    def step(self, dev, stamp, poison):
        """Called once per eval() forward, after the cached copies are current for `stamp` (the tuple of the layers' cache
        keys): records the fingerprints when the copies were just (re)built, otherwise checks them.  Returns True when the
        PREVIOUS check had found a difference (the caller invalidates, rebuilds and calls again)."""
        # (host cost matters: a step that reads the host once -- an MD step with an exact neighbour list -- pays every
        # microsecond spent here in full; the parameter table is only rebuilt when the caches were)
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing and self.tripped():
            self._event = None
            self.armed_for = None
            warnings.warn("hermnet_amd: parameters were modified behind the cached kernel-ready copies (a write through "
                          "`.data`?): the previous result was NaN by design; the copies are rebuilt now.  Call "
                          "model.invalidate_caches() after such writes.", RuntimeWarning, stacklevel=3)
            return True
        if self.armed_for != stamp:
            if self._table(dev) == 0:
                return False
            self.flag.zero_()
            self._launch(False, None)
            self.armed_for = stamp
            return False
        self._launch(True, poison)
        if not capturing:
            self._flag_host.copy_(self.flag, non_blocking=True)
            if self._event is None:
                self._event = torch.cuda.Event()
            self._event.record()
        return False
