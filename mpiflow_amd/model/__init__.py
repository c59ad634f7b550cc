"""MPI producer: the AdaMPI network that emits the [B,S,4,H,W] plane stack the render path consumes.

Own module definitions, state-dict compatible with the reference's checkpoints ({'num_planes', 'weight'},
gen_3dphoto_dynamic_v2.py:52-58).  Dense convolutions run on stock PyTorch-ROCm / MIOpen - they are not a kernel target of
this build; what is specific here is the output contract (channel-planar rgb = sigmoid, sigma = relu(x * cum_mask) + 1e-4)
and the optional hand-off of the *raw* last-layer output so that the activation epilogue is fused into Stage A+C."""
from .adampi import MPIPredictor  # noqa: F401
