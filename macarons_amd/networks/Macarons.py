"""Host-side mirror of the SCONE part of macarons/networks/Macarons.py: `Macarons.forward(mode=...)` dispatch
(:110-136) and `compute_visibility_gains` (:138-178).  The depth module (ManyDepth) is out of scope for this
path: any nn.Module can be passed as `depth_model` and is called as in the reference.
"""
from torch import nn

from .. import ops


class Macarons(nn.Module):
    def __init__(self, depth_model, occupancy_model, visibility_model):
        super().__init__()
        self.depth = depth_model
        self.occupancy = occupancy_model
        self.visibility = visibility_model
        if depth_model is not None:
            self.image_height = depth_model.input_height
            self.image_width = depth_model.input_width

    def forward(self, mode, x=None, x_alpha=None, R=None, T=None, zfar=None, device=None, gt_pose=None,
                partial_point_cloud=None, proxy_points=None, view_harmonics=None):
        if mode == 'depth':
            if (x is None) or (x_alpha is None) or (R is None) or (T is None) or (zfar is None) or (device is None):
                raise NameError("For 'occupancy' mode, you should provide the following args:"
                                "x, x_alpha, R, T, zfar, device")                      # message as in Macarons.py:117
            return self.depth(x=x, x_alpha=x_alpha, R=R, T=T, zfar=zfar, device=device, gt_pose=gt_pose)
        elif mode == 'occupancy':
            if (partial_point_cloud is None) or (proxy_points is None) or (view_harmonics is None):
                raise NameError("For 'occupancy' mode, you should provide the following args:"
                                "partial_point_cloud, proxy_points, view_harmonics")
            return self.occupancy(pc=partial_point_cloud, x=proxy_points, view_harmonics=view_harmonics)
        elif mode == 'visibility':
            if (proxy_points is None) or (view_harmonics is None):
                raise NameError("For 'visibility' mode, you should provide the following args:"
                                "proxy_points, view_harmonics")
            return self.visibility(proxy_points, view_harmonics=view_harmonics)
        raise NameError("Invalid mode. Please select a mode between 'depth', 'occupancy' and 'visibility'.")

    def compute_visibility_gains(self, pts, harmonics, X_cam):
        """-> [n_clouds, n_camera_candidates, seq_len]  (Macarons.py:138-178; raises for ReLU like :175-176)."""
        if not self.visibility.use_sigmoid:
            raise NameError("WARNING! ReLU has been used in visibility model.")
        return self.visibility.compute_visibilities(pts, harmonics, X_cam)
