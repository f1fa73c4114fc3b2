"""The switches of the host code, in one place.

Environment variables (read at call time) -- the ones a USER of the library may need:
  HERMNET_LIB_PATH         another build of libhermnet_hip.so (hermnet_amd/_lib.py; with HERMNET_ALLOW_STALE_LIB=1 its source
                           stamp is not checked)
  HERMNET_PARAM_GUARD=0    no device-side fingerprint of the parameters behind the cached kernel-ready copies (guard.py)
  HERMNET_HALO_OVERLAP     atom shards, the per-layer halo exchange: 2 (default) inside the consuming layer in its "proj" form
                           (projected rows forward, partial sums of gradients backward), 1 the round-4 form (x | vec rows around
                           windowed node launches), 0 the blocking exchange in front of the layer
  HERMNET_DEFER_SUMS=0     finishing launches instead of handing a layer's input gradients down as partial sums
  HERMNET_DEBUG_POISON=1   NaN in every buffer that a later launch must fill (halo rows until they are unpacked, the
                           edge-gradient sink, gradients handed down): what the bit-for-bit tests run with

Everything else that used to be an environment A/B knob was decided by measurement and is gone.  The alternative
FORMS that tests compare the default against are plain attributes below -- tests monkeypatch them, nothing reads them from the
environment; kernel-side variants are library options (`_lib.set_option`, include/hermnet_hip.h: HN_OPT_*)."""
import os

fused_layer = True          # False: the layer's node algebra through PyTorch autograd over the same edge kernels (debug path)
node_chain = True           # False: library GEMMs + stage kernels instead of the node chain kernels (the path of widths > 512)
boundary_mode = 0           # layer.FusedRelationalLayer: 0 every phase a launch of its own; 4 / 1 / 3 / 2: the fused 16-row forms
native_relations = True     # False: the relation build as torch ops (its definition: relations.py) instead of the HIP kernels
train_row_sums = True       # False: train(): per-edge message kernels + segmented sums instead of row sums inside the kernels


def _env(name, default):
    return os.environ.get("HERMNET_" + name, default)


def debug_poison():
    return _env("DEBUG_POISON", "0") != "0"


def halo_overlap():
    return _env("HALO_OVERLAP", "2")


def defer_sums():
    return _env("DEFER_SUMS", "1") != "0"


def param_guard():
    return _env("PARAM_GUARD", "1") != "0"
