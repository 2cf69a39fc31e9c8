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

    # mode -> (sub-module attribute, required keyword arguments, how the sub-module takes them, label used in the error text).
    # The reference's message for a missing 'depth' argument says 'occupancy' (Macarons.py:117); kept, callers match on the text.
    _MODES = {
        "depth": ("depth", ("x", "x_alpha", "R", "T", "zfar", "device"), "occupancy",
                  lambda m, a: m(x=a["x"], x_alpha=a["x_alpha"], R=a["R"], T=a["T"], zfar=a["zfar"], device=a["device"],
                                 gt_pose=a["gt_pose"])),
        "occupancy": ("occupancy", ("partial_point_cloud", "proxy_points", "view_harmonics"), "occupancy",
                      lambda m, a: m(pc=a["partial_point_cloud"], x=a["proxy_points"], view_harmonics=a["view_harmonics"])),
        "visibility": ("visibility", ("proxy_points", "view_harmonics"), "visibility",
                       lambda m, a: m(a["proxy_points"], view_harmonics=a["view_harmonics"])),
    }

    def forward(self, mode, x=None, x_alpha=None, R=None, T=None, zfar=None, device=None, gt_pose=None,
                partial_point_cloud=None, proxy_points=None, view_harmonics=None):
        """Macarons.py:110-136 (same signature: the first group of arguments belongs to the depth module, the second to SCONE)."""
        args = dict(x=x, x_alpha=x_alpha, R=R, T=T, zfar=zfar, device=device, gt_pose=gt_pose,
                    partial_point_cloud=partial_point_cloud, proxy_points=proxy_points, view_harmonics=view_harmonics)
        entry = self._MODES.get(mode)
        if entry is None:
            raise NameError("Invalid mode. Please select a mode between " + ", ".join(f"'{k}'" for k in list(self._MODES)[:-1])
                            + f" and '{list(self._MODES)[-1]}'.")
        attr, required, label, call = entry
        if any(args[k] is None for k in required):
            raise NameError(f"For '{label}' mode, you should provide the following args:" + ", ".join(required))
        return call(getattr(self, attr), args)

    def compute_visibility_gains(self, pts, harmonics, X_cam):
        """-> [n_clouds, n_camera_candidates, seq_len]  (Macarons.py:138-178; raises for ReLU like :175-176)."""
        if not self.visibility.use_sigmoid:
            raise NameError("WARNING! ReLU has been used in visibility model.")
        return self.visibility.compute_visibilities(pts, harmonics, X_cam)
