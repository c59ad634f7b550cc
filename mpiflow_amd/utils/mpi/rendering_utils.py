"""Drop-in for the on-path part of the reference's utils/mpi/rendering_utils.py (transform_G_xyz, :4-23).

The training-time samplers of that file (gather_pixel_by_pxpy, uniformly_sample_disparity_*, sample_pdf, :26-139) are
never called on the generation path and are not provided."""
import torch

from ... import ops


def transform_G_xyz(G, xyz, is_return_homo=False):
    """xyz_out = (G . [xyz; 1])[:3]  -  reference utils/mpi/rendering_utils.py:4-23.

    :param G: Bx4x4 (or 4x4 with xyz 3xN); every batch entry must hold the same matrix (the reference's callers
              repeat one G over the S planes, utils/mpi/mpi_rendering.py:251-254)
    :param xyz: Bx3xN on the GPU
    :return: Bx3xN (Bx4xN with a row of ones when is_return_homo)"""
    assert len(G.size()) == len(xyz.size())
    squeeze = len(G.size()) == 2
    G_B44 = G.unsqueeze(0) if squeeze else G
    xyz_B3N = xyz.unsqueeze(0) if squeeze else xyz
    G_cpu = G_B44.detach().to("cpu", torch.float32)
    if not bool((G_cpu == G_cpu[0:1]).all()):
        out = torch.stack([ops.transform_xyz(G_cpu[b], xyz_B3N[b:b + 1])[0] for b in range(G_cpu.shape[0])])
    else:
        out = ops.transform_xyz(G_cpu[0], xyz_B3N)
    if is_return_homo:
        out = torch.cat((out, torch.ones_like(out[:, 0:1, :])), dim=1)
    return out[0] if squeeze else out
