"""MPI producer: the AdaMPI network that emits the [B,S,4,H,W] plane stack the render path consumes.

`MPIPredictor` (adampi.py): own module definitions, state-dict compatible with the reference's checkpoints
({'num_planes', 'weight'}, gen_3dphoto_dynamic_v2.py:52-58); plain torch, fp32 - the bit-exact mirror of the reference
network.  `HipPredictor` (engine.py): the same network with its 20 per-plane convolutions on the MFMA convolution engine of
libmpiflow_hip.so (fp16 storage, fp32 accumulation), replayed from one hipGraph.  Both can hand the decoder's *raw* output to
Stage A+C, which applies the activation epilogue (rgb = sigmoid, sigma = relu(x * cum_mask) + 1e-4) in registers."""
from .adampi import MPIPredictor  # noqa: F401


def __getattr__(name):                      # HipPredictor binds the HIP library: import it only when asked for
    if name == "HipPredictor":
        from .engine import HipPredictor
        return HipPredictor
    raise AttributeError(name)
