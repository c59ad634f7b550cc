"""Drop-in for the reference's geometry.py (monodepth2 heritage): BackprojectDepth, Project3D,
transformation_from_parameters, get_translation_matrix, rot_from_axisangle.

Per-pixel work (back-projection and projection of h*w points) runs in HIP (mpf_backproject / mpf_project3d); the 4x4
pose algebra stays in torch, as in the reference, on whatever device its inputs are."""
import torch
import torch.nn as nn

from . import host_math, ops

__all__ = ["BackprojectDepth", "Project3D", "transformation_from_parameters"]

transformation_from_parameters = host_math.transformation_from_parameters   # geometry.py:79-95
get_translation_matrix = host_math.get_translation_matrix                    # geometry.py:98-111
rot_from_axisangle = host_math.rot_from_axisangle                            # geometry.py:114-153


class BackprojectDepth(nn.Module):
    """Depth image -> point cloud (reference geometry.py:17-49)."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        assert batch_size == 1, "the path is batch-1 (moving_obj.py:37)"
        self.batch_size, self.height, self.width = batch_size, height, width

    def forward(self, depth, inv_K):
        """cam_points [B,4,h*w] = (depth * (inv_K[:3,:3] . (x, y, 1)) ; 1)   (:41-49)"""
        H, W = self.height, self.width
        ik = inv_K.detach().to("cpu", torch.float32).reshape(-1, inv_K.shape[-2], inv_K.shape[-1])[0, :3, :3]
        return ops.backproject(depth.reshape(H, W), ik).unsqueeze(0)


class Project3D(nn.Module):
    """3D points -> pixel coordinates in camera (K, T) (reference geometry.py:52-76)."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        assert batch_size == 1
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T, T2=None):
        """:return: (pix_coords [B,h,w,2] normalised to [-1,1], z [B,1,h*w])   (:63-76)"""
        K = K.detach().to("cpu", torch.float32)
        T = T.detach().to("cpu", torch.float32)
        if T2 is not None:
            T = torch.matmul(T, torch.inverse(T2.detach().to("cpu", torch.float32)))
        P = torch.matmul(K, T)[:, :3, :]
        H, W = self.height, self.width
        pix, z = ops.project3d(points[0], P[0], H, W, eps=self.eps)
        return pix.unsqueeze(0), z.reshape(1, 1, H * W)
