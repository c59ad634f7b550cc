"""Drop-in for the reference's utils/mpi/homography_sampler.py: same class, same method signatures, same return
values; the per-pixel work runs in HIP kernels (mpf_homography_sample / mpf_homography_flow), the [B,3,3] homographies
are built on the host exactly as the reference builds them (batched torch-CPU matmuls + one fp64 inverse)."""
import numpy as np
import torch

from ... import host_math, ops


def inverse(matrices):
    """Batched inverse, retried up to 5 times on NaN, then Exception("Matrix inverse contains nan!")
    (reference utils/mpi/homography_sampler.py:6-27)."""
    return host_math.inverse(matrices)


class HomographySample:
    """reference utils/mpi/homography_sampler.py:30-220"""

    def __init__(self, H_tgt, W_tgt, device=None):
        self.device = torch.device("cpu") if device is None else device
        self.Height_tgt = H_tgt
        self.Width_tgt = W_tgt
        self.meshgrid = self.grid_generation(self.Height_tgt, self.Width_tgt, self.device)
        self.meshgrid = self.meshgrid.permute(2, 0, 1).contiguous()  # 3xHxW
        self.n = self.plane_normal_generation(self.device)

    @staticmethod
    def grid_generation(H, W, device):
        """HxWx3 grid of (x, y, 1), x in [0, W-1]   (:46-56)"""
        x = np.linspace(0, W - 1, W)
        y = np.linspace(0, H - 1, H)
        xv, yv = np.meshgrid(x, y)
        xv = torch.from_numpy(xv.astype(np.float32)).to(dtype=torch.float32, device=device)
        yv = torch.from_numpy(yv.astype(np.float32)).to(dtype=torch.float32, device=device)
        return torch.stack((xv, yv, torch.ones_like(xv)), dim=2)

    @staticmethod
    def plane_normal_generation(device):
        return torch.tensor([0, 0, 1], dtype=torch.float32, device=device)

    @staticmethod
    def euler_to_rotation_matrix(x_angle, y_angle, z_angle, seq='xyz', degrees=False):
        """Unused on the path (reference :64-78); kept for signature parity."""
        from scipy.spatial.transform import Rotation
        r = Rotation.from_euler(seq, [-x_angle, -y_angle, -z_angle], degrees=degrees)
        return r.as_matrix().astype(np.float32)

    @staticmethod
    def _homographies(d_src_B, G_tgt_src, K_src_inv, K_tgt):
        """H_tgt_src [B,3,3], as :105-118, on the host in fp32."""
        B = d_src_B.reshape(-1).shape[0]
        G = G_tgt_src.detach().to("cpu", torch.float32)
        Kinv = K_src_inv.detach().to("cpu", torch.float32)
        Kt = K_tgt.detach().to("cpu", torch.float32)
        d = d_src_B.detach().to("cpu", torch.float32)
        R = G[:, 0:3, 0:3]
        t = G[:, 0:3, 3]
        n = torch.tensor([0, 0, 1], dtype=torch.float32).unsqueeze(0).repeat(B, 1)
        d33 = d.reshape(B, 1, 1).repeat(1, 3, 3)
        R_tnd = R - torch.matmul(t.unsqueeze(2), n.unsqueeze(1)) / -d33
        return torch.matmul(Kt, torch.matmul(R_tnd, Kinv))

    def sample(self, src_BCHW, d_src_B, G_tgt_src, K_src_inv, K_tgt):
        """Warp every plane b of src_BCHW to the target view by its plane-induced homography (:80-158).
        :return: tgt_BCHW, valid_mask BxHxW (bool), flowB2A BxHxWx2"""
        assert src_BCHW.size(2) == self.Height_tgt and src_BCHW.size(3) == self.Width_tgt, \
            "the HIP sampler renders at the source resolution (as every caller in the reference does)"
        H_tgt_src = self._homographies(d_src_B, G_tgt_src, K_src_inv, K_tgt)
        with torch.no_grad():
            H_src_tgt = inverse(H_tgt_src.to(torch.float64)).to(torch.float32)
        tgt, valid, flow = ops.homography_sample(src_BCHW, H_src_tgt)
        return tgt.to(src_BCHW.dtype), valid, flow

    def sample_inverse(self, src_BCHW, d_src_B, G_tgt_src, K_src_inv, K_tgt):
        """Per-plane forward flow of every source pixel (:160-220).  src_BCHW is used for its shape/device only.
        :return: flowA2B BxHxWx2"""
        H_tgt_src = self._homographies(d_src_B, G_tgt_src, K_src_inv, K_tgt)
        return ops.homography_flow(H_tgt_src, self.Height_tgt, self.Width_tgt, src_BCHW.device)
