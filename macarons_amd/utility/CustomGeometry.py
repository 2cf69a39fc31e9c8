"""Mirror of macarons/utility/CustomGeometry.py:5-45 (host-side helpers on plain torch tensors)."""
import numpy as np
import torch


def get_cartesian_coords(r, elev, azim, in_degrees=False):
    factor = np.pi / 180. if in_degrees else 1
    X = torch.stack((torch.cos(factor * elev) * torch.sin(factor * azim), torch.sin(factor * elev),
                     torch.cos(factor * elev) * torch.cos(factor * azim)), dim=2)
    return r * X.view(-1, 3)


def get_spherical_coords(X):
    """(r, elev, azim): Y-up, azimuth from +Z toward +X (CustomGeometry.py:27-45), without the reference's
    boolean-mask scatters (torch.where keeps it sync-free)."""
    r_x = torch.linalg.norm(X, dim=1)
    yr = X[:, 1] / r_x
    elev_x = torch.asin(yr)
    elev_x = torch.where(yr <= -1, torch.full_like(elev_x, -np.pi / 2), elev_x)
    elev_x = torch.where(yr >= 1, torch.full_like(elev_x, np.pi / 2), elev_x)
    q = X[:, 2] / (r_x * torch.cos(elev_x))
    azim_x = torch.acos(q)
    azim_x = torch.where(q <= -1, torch.full_like(azim_x, np.pi), azim_x)
    azim_x = torch.where(q >= 1, torch.zeros_like(azim_x), azim_x)
    azim_x = torch.where(X[:, 0] < 0, -azim_x, azim_x)
    return r_x, elev_x, azim_x
