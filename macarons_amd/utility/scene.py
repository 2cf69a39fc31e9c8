"""A uniform grid of capped point sets: the subset of macarons/utility/macarons_utils.py `Scene` (:2588-2830) and `Cell`
(:2503-2585) that the occupancy-field pass walks -- cell lookup, 27-neighbourhoods, per-cell point / feature
stores with the admission test of Cell.fill on the MI355X (fp64 nearest-distance kernel), proxy-point state tensors.  Host-side
bookkeeping in the reference's own dict-of-cells shape (keys like '[i, j, k]'), so that
macarons_amd.utility.macarons_utils.compute_scene_occupancy_probability_field accepts either these objects or the reference's.
Everything else of the reference classes (rendering, coverage metrics, collision tests, memory) is out of scope.
"""
import math

import numpy as np
import torch

from . import macarons_utils as mu


def _key(idx):
    return str([int(v) for v in idx])


class Cell:
    """One grid cell: axis-aligned box + the points (and optional per-point features) kept in it."""

    def __init__(self, center, l, w, h, capacity, resolution, device, feature_dim=0):
        half = torch.tensor([[float(l) / 2., float(w) / 2., float(h) / 2.]], device=center.device)
        self.center, self.l, self.w, self.h = center, l, w, h
        self.x_min, self.x_max = center - half, center + half
        if resolution is None and capacity is None:
            raise NameError("Please choose a capacity or a resolution.")
        # the largest cross-section of the box sets how many discs of diameter `resolution` fit (macarons_utils.py:2515-2541)
        l_, w_, h_ = float(l), float(w), float(h)
        area = max(l_ * math.hypot(w_, h_), w_ * math.hypot(h_, l_), h_ * math.hypot(l_, w_))
        if resolution is None:
            resolution = 2 * math.sqrt(area / capacity / math.pi)
        elif capacity is None:
            capacity = int(area // (math.pi * (resolution / 2.) ** 2))
        self.capacity, self.resolution, self.device = capacity, resolution, device
        self.use_feature, self.feature_dim = feature_dim > 0, feature_dim
        self.empty()

    # ---- storage: a cell of a Scene is a SEGMENT of the scene's flat store (all cells' rows, cells in linear order) whenever that
    # store is current; assigning to a cell's tensors hands every cell its own tensors back (Scene._release_store)
    _scene, _lin, _pts, _fts = None, -1, None, None

    def _served(self):
        return self._scene is not None and self._scene._store is not None

    @property
    def cell_pts(self):
        if self._served():
            st = self._scene._store
            return st.pts[int(st.off[self._lin]):int(st.off[self._lin + 1])]
        return self._pts

    @cell_pts.setter
    def cell_pts(self, v):
        if self._served():
            self._scene._release_store()
        self._pts = v

    @property
    def cell_features(self):
        if self._served():
            st = self._scene._store
            return st.fts[int(st.off[self._lin]):int(st.off[self._lin + 1])]
        return self._fts

    @cell_features.setter
    def cell_features(self, v):
        if self._served():
            self._scene._release_store()
        self._fts = v

    def n_points(self):
        if self._served():
            st = self._scene._store
            return int(st.off[self._lin + 1] - st.off[self._lin])
        return int(self._pts.shape[0])

    def empty(self):
        self.cell_pts = torch.zeros(0, 3, device=self.device)
        if self.use_feature:
            self.cell_features = torch.zeros(0, self.feature_dim, device=self.device)

    def is_empty(self):
        return self.cell_pts.shape[0] == 0

    def fill(self, pts, features=None, n_point_min=0, perm=None):
        """Cell.fill (:2551-2577): keep the points strictly inside the box whose fp64 distance to every stored point exceeds
        the resolution, append, then keep a random `capacity` subset (torch.randperm on the CPU generator, or `perm`)."""
        inside = (torch.max(pts - self.x_max, dim=-1)[0] < 0.) & (torch.min(pts - self.x_min, dim=-1)[0] > 0.)
        add = pts[inside]
        if add.shape[0] <= n_point_min:                 # (the reference tests the two faces one after the other; same outcome)
            return
        fts = features[inside] if (self.use_feature and features is not None) else None
        if self.cell_pts.shape[0] > 0:
            keep = mu.cell_fill_mask(add.contiguous(), self.cell_pts.contiguous(), self.resolution)
            add = add[keep]
            fts = fts[keep] if fts is not None else None
        self.cell_pts = torch.vstack((self.cell_pts, add))
        idx = (torch.randperm(len(self.cell_pts)) if perm is None else perm)[:self.capacity].to(self.cell_pts.device)
        self.cell_pts = self.cell_pts[idx]
        if fts is not None:
            self.cell_features = torch.vstack((self.cell_features, fts))[idx]


class _Store:
    """Flat store of a Scene: every cell's points (and features), cells in linear-id order.  off: host int64 [n_cells + 1]; off_dev: the
    same on the device."""
    __slots__ = ("pts", "fts", "off", "off_dev")

    def __init__(self, pts, fts, off, off_dev):
        self.pts, self.fts, self.off, self.off_dev = pts, fts, off, off_dev


class Scene:
    def __init__(self, x_min, x_max, grid_l, grid_w, grid_h, cell_capacity, cell_resolution, n_proxy_points, device,
                 view_state_n_elev=7, view_state_n_azim=2 * 7, feature_dim=0, score_threshold=1.):
        self.grid_l, self.grid_w, self.grid_h = grid_l, grid_w, grid_h
        self.x_min, self.x_max = 0. + x_min, 0. + x_max
        ext = self.x_max - self.x_min
        self.l, self.w, self.h = ext[0] / grid_l, ext[1] / grid_w, ext[2] / grid_h
        self.device, self.feature_dim = device, feature_dim
        self.cells = {}
        self._store = None
        for i in range(grid_l):
            for j in range(grid_w):
                for k in range(grid_h):
                    center = torch.Tensor([self.x_min[0] + (0.5 + i) * self.l, self.x_min[1] + (0.5 + j) * self.w,
                                           self.x_min[2] + (0.5 + k) * self.h]).to(device)
                    cell = Cell(center, self.l, self.w, self.h, cell_capacity, cell_resolution, device, feature_dim)
                    cell_capacity, cell_resolution = cell.capacity, cell.resolution      # derived once, shared by all cells
                    cell._scene, cell._lin = self, (i * grid_w + j) * grid_h + k
                    self.cells[_key((i, j, k))] = cell
        self.cell_capacity, self.cell_resolution = cell_capacity, cell_resolution
        self.n_proxy_points = n_proxy_points
        self.view_state_n_elev, self.view_state_n_azim = view_state_n_elev, view_state_n_azim
        self.n_view_state_cameras = view_state_n_elev * view_state_n_azim
        self.score_threshold = score_threshold
        self.proxy_points = self.proxy_proba = self.proxy_supervision_occ = self.view_states = self.out_of_field = None
        # the volume per proxy point is an fp32 tensor division upstream (:2670-2674), the cube root a float64 numpy one
        vol = ((self.l * self.w * self.h) / (n_proxy_points / (grid_l * grid_h * grid_w))).item()
        self.distance_between_proxy_points = 2 * np.power(3 * vol / (4 * np.pi), 1. / 3.)

    # ---- cell lookup ----
    def _consts(self, device):
        """Small per-device constants of the grid (built once: creating them from Python lists on every call is a blocking
        host-to-device copy each)."""
        c = getattr(self, "_dev_consts", None)
        if c is None or c["device"] != str(device):
            c = {"device": str(device),
                 "step": torch.stack((self.l, self.w, self.h)).to(device).view(1, 3),
                 "x_min": self.x_min.to(device), "x_max": self.x_max.to(device),
                 "hi": torch.tensor([self.grid_l - 1, self.grid_w - 1, self.grid_h - 1], device=device, dtype=torch.float32),
                 "gc": torch.cat((self.x_min.reshape(3).float(), self.x_max.reshape(3).float(),
                                  torch.stack((self.l, self.w, self.h)).reshape(3).float())).to(device).contiguous(),
                 "lin": torch.tensor([self.grid_w * self.grid_h, self.grid_h, 1], device=device)}
            self._dev_consts = c
        return c

    def get_cells_for_each_pt(self, pts):
        c = self._consts(pts.device)
        step = c["step"]
        d = pts - c["x_min"]
        idx = (d - d % step) / step                                   # utils.floor_divide (non-negative modulo)
        return torch.minimum(idx, c["hi"].to(idx.dtype)).long().clamp_(min=0)

    def linear_cell_ids(self, pts):
        """int32 [N]: (i_l * grid_w + i_w) * grid_h + i_h of get_cells_for_each_pt(pts), one fused launch on a HIP device."""
        if pts.is_cuda:
            from .. import ops
            return ops.cell_keys(pts, self._consts(pts.device)["gc"], (self.grid_l, self.grid_w, self.grid_h))
        return (self.get_cells_for_each_pt(pts) * self._consts(pts.device)["lin"]).sum(-1).to(torch.int32)

    def get_englobing_cells(self, pts, list=False):
        res = torch.unique(self.get_cells_for_each_pt(pts), dim=0)
        return res.cpu().numpy().tolist() if list else res

    def get_neighboring_cells(self, cell_idx):
        shift = torch.cartesian_prod(torch.arange(3), torch.arange(3), torch.arange(3)).to(cell_idx.device) - 1
        hi = torch.tensor([self.grid_l - 1, self.grid_w - 1, self.grid_h - 1], device=cell_idx.device)
        return torch.unique(torch.minimum((cell_idx + shift).clamp(min=0), hi), dim=0)

    def get_key_from_idx(self, cell_idx):
        return _key(cell_idx.cpu().numpy().tolist())

    # ---- stores ----
    def get_pts_in_bounding_box(self, pts, return_mask=True):
        m = ((pts >= self.x_min.to(pts.device)) & (pts <= self.x_max.to(pts.device))).all(dim=-1)
        return (pts[m], m) if return_mask else pts[m]

    def _cell_table(self):
        """Cells in linear-id (= lexicographic key) order and their box bounds as [n_cells, 3] device tables."""
        t = getattr(self, "_table", None)
        if t is None:
            order = [self.cells[_key((i, j, k))] for i in range(self.grid_l) for j in range(self.grid_w) for k in range(self.grid_h)]
            lo = torch.stack([c.x_min.reshape(3) for c in order]).to(self.device)
            hi = torch.stack([c.x_max.reshape(3) for c in order]).to(self.device)
            t = self._table = (order, lo, hi)
        return t

    # ---- the flat store ----
    def flat_store(self):
        """The cells' points / features as ONE pair of tensors (cells in linear order) + offsets; built from the cells' own tensors when
        something assigned to them since the last fill."""
        if self._store is None:
            from .. import ops
            cells, _, _ = self._cell_table()
            lens = [int(c._pts.shape[0]) for c in cells]
            off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
            dev = self.device
            pts = torch.cat([c._pts for c, n in zip(cells, lens) if n] + [torch.zeros(0, 3, device=dev)]).contiguous()
            fts = None
            if self.feature_dim > 0:
                # one feature row per stored point: upstream's Cell.fill(features=None) leaves cell_features behind the points, which
                # would silently shift every later cell's rows in the flat store -- refuse instead of mis-aligning
                rows = []
                for c, n in zip(cells, lens):
                    if n:
                        f_ = None if c._fts is None else c._fts.reshape(-1, self.feature_dim)
                        if f_ is None or f_.shape[0] != n:
                            raise ValueError(f"Scene.flat_store: a cell holds {n} points but {0 if f_ is None else f_.shape[0]} feature rows "
                                             f"(feature_dim={self.feature_dim}): fill cells with `features=` of one row per point")
                        rows.append(f_.to(torch.float32))
                fts = torch.cat(rows + [torch.zeros(0, self.feature_dim, device=dev)]).contiguous()
            self._store = _Store(pts, fts, off, ops.h2d(off, torch.int64, dev))
            for c in cells:
                c._pts = c._fts = None
        return self._store

    def _release_store(self):
        """Every cell takes its segment as a tensor of its own (somebody is about to assign to a cell)."""
        st, self._store = self._store, None
        if st is None:
            return
        for c in self._cell_table()[0]:
            o0, o1 = int(st.off[c._lin]), int(st.off[c._lin + 1])
            c._pts = st.pts[o0:o1]
            if st.fts is not None:
                c._fts = st.fts[o0:o1]

    def fill_cells_begin(self, pts, features=None, n_point_min=0, valid=None):
        """Device part of fill_cells (ops.scene_fill_begin): nothing returns to the host.  -> handle for fill_cells_end; its `.counts`
        (device int64) holds, per cell, the number of candidates and of admitted candidates."""
        from .. import ops
        self._check_grid_size()
        cells, lo, hi = self._cell_table()
        st = self.flat_store()
        with_fts = self.feature_dim > 0 and features is not None
        h = ops.scene_fill_begin(pts, valid, self._consts(self.device)["gc"], (self.grid_l, self.grid_w, self.grid_h), lo, hi, st.pts, st.off_dev,
                                 cells[0].resolution, n_point_min, features.reshape(pts.shape[0], self.feature_dim) if with_fts else None)
        return h

    MAX_CELLS = 1023        # the counting sort of csrc/scene.hip (grp_* kernels: one LDS counter per cell) covers grids of < 1024 cells

    def _check_grid_size(self):
        n = self.grid_l * self.grid_w * self.grid_h
        if n > self.MAX_CELLS:
            raise NotImplementedError(f"Scene: the fused fill / field passes handle grids of up to {self.MAX_CELLS} cells, this one has {n} "
                                      f"({self.grid_l} x {self.grid_w} x {self.grid_h}); the reference's scenes use 18 .. 72")

    def fill_counts(self, host_counts):
        """(candidates per cell, admitted per cell) out of a host copy of a fill handle's counts."""
        nk = self.grid_l * self.grid_w * self.grid_h
        return host_counts[:nk], host_counts[2 * nk + 3:3 * nk + 3]

    def fill_overflows(self, cand, adm, n_point_min=0):
        """True when a touched cell would exceed its capacity: the subset Cell.fill keeps is then random (the draws decide WHICH points
        stay), otherwise only their order is."""
        st = self.flat_store()
        cap = self._cell_table()[0][0].capacity
        b_len = np.diff(st.off)
        return bool(np.any((cand > n_point_min) & (b_len + adm > cap)))

    def fill_cells_draw(self, h, cand, adm, n_point_min=0, group=None):
        """Host part of fill_cells, first half: the touched cells' torch.randperm draws on the CPU generator in cell order (Cell.fill
        :2573 -- the reference's draws).  -> plan for fill_cells_apply (None: no cell was touched).  Nothing is launched."""
        from .. import dist as mdist
        cells, _, _ = self._cell_table()
        st = self.flat_store()
        cap = cells[0].capacity
        b_off = st.off
        b_len = np.diff(b_off)
        adm = np.asarray(adm, np.int64)
        touched = np.asarray(cand) > n_point_min                # Cell.fill returns before the random subset (:2562): no draw
        if not touched.any():
            return None
        n_comb = b_len + adm
        n_keep = np.where(touched, np.minimum(n_comb, cap), b_len)
        world, rank_ = mdist.group_world_rank(group)           # group=None: local, whatever process groups exist
        plan = {"h": h, "touched": touched, "n_keep": n_keep, "adm": adm, "b_len": b_len, "group": group, "world": world,
                "pm": None}
        if rank_ == 0:
            # torch.randperm(n_comb)[:capacity] per touched cell, cell order, CPU generator (:2573) -- the reference's draws, made by one
            # call into the C++ extension (the same at::randperm calls without ~70 dispatcher round trips)
            from .host import batched_draws_ok
            t_idx = np.nonzero(touched)[0]
            if batched_draws_ok():
                from .. import torch_ops  # noqa: F401
                plan["pm"] = torch.ops.macarons.randperm_prefixes([int(n_comb[c]) for c in t_idx], [int(cap)] * len(t_idx)).numpy()
            else:                                                       # torch.randperm was replaced from Python: honour it
                plan["pm"] = np.concatenate([torch.randperm(int(n_comb[c]))[:cap].numpy() for c in t_idx])
        return plan

    def fill_cells_apply(self, plan):
        """Host part of fill_cells, second half: the draws become ONE gather that writes the scene's new flat store."""
        from .. import ops
        from .. import dist as mdist
        if plan is None:
            return
        h, touched, n_keep, adm, b_len = plan["h"], plan["touched"], plan["n_keep"], plan["adm"], plan["b_len"]
        group, world = plan["group"], plan["world"]
        cells, _, _ = self._cell_table()
        n_cells, dev = len(cells), self.device
        st = self.flat_store()
        b_off = st.off
        n_store = int(b_off[-1])
        adm_off = np.concatenate(([0], np.cumsum(adm))).astype(np.int64)
        new_off = np.concatenate(([0], np.cumsum(n_keep))).astype(np.int64)
        n_new = int(new_off[-1])
        F = self.feature_dim
        # ONE upload: the cells' tables and the permutation prefixes; the gather maps every new row to its source on the device
        n_pm_cell = np.where(touched, n_keep, 0)
        pm_off = np.concatenate(([0], np.cumsum(n_pm_cell))).astype(np.int64)
        n_pm = int(pm_off[-1])
        pm = plan["pm"] if plan["pm"] is not None else np.zeros(n_pm, np.int64)
        tabs = np.concatenate([new_off, b_off, adm_off, pm_off, np.concatenate((touched.astype(np.int64), [0]))])
        pm32 = np.ascontiguousarray(pm, dtype=np.int32)
        if n_pm % 2:
            pm32 = np.concatenate((pm32, np.zeros(1, np.int32)))
        buf = ops.h2d(np.concatenate((tabs, pm32.view(np.int64))), torch.int64, dev)
        if mdist.exchange_on(group):                        # rank 0's draws for every replica
            mdist.broadcast(buf, 0, group)
        new_pts, new_fts = ops.scene_fill_gather_perm(buf, n_pm, n_cells, n_new, h, st.pts, st.fts, n_store, F)
        from .. import ops as _o
        self._store = _Store(new_pts, new_fts, new_off, _o.h2d(new_off, torch.int64, dev))

    def fill_cells_end(self, h, cand, adm, n_point_min=0, group=None):
        """Host part of fill_cells (fill_cells_draw + fill_cells_apply)."""
        self.fill_cells_apply(self.fill_cells_draw(h, cand, adm, n_point_min, group))

    def fill_cells(self, pts, features=None, n_point_min=0, group=None, valid=None):
        """Scene.fill_cells (macarons_utils.py:2727-2737) over Cell.fill (:2551-2577) for ALL touched cells at once: upstream loops
        the cells from Python, each testing every point against its box and its store.  Here (fill_cells_begin) a counting sort groups
        the points by cell (floor rule, then the strict box test of that cell), ONE segmented fp64 nearest-distance launch runs every
        cell's admission test against its own store, a second grouping compacts the admitted points, and (fill_cells_end) the host --
        after reading two integers per cell -- draws each touched cell's torch.randperm on the CPU generator in cell order (the
        reference's draws) and turns them into one gather.  A point exactly on a cell face belongs to no cell, as upstream.
        `group` (a torch.distributed group whose ranks hold replicas of this scene and call together; None = a local call whatever
        process groups exist): the permutations are rank 0's, broadcast once -- every rank drawing its own would let the replicas
        diverge.
        `valid` (bool [N], optional): only these rows of pts are offered -- the same as fill_cells(pts[valid], features[valid]) without
        the read-back that boolean indexing costs."""
        if pts.shape[0] == 0:
            return
        h = self.fill_cells_begin(pts, features, n_point_min, valid)
        cand, adm = self.fill_counts(h.counts.cpu().numpy())                                       # the one read-back
        self.fill_cells_end(h, cand, adm, n_point_min, group)

    def get_pt_cloud_from_cells(self, cell_indices, return_features=True):
        with_fts = return_features and self.feature_dim > 0
        keys = ([_key(c) for c in cell_indices.cpu().numpy().tolist()] if cell_indices.dim() > 1
                else [_key(cell_indices.cpu().numpy().tolist())])
        pts = torch.vstack([torch.zeros(0, 3, device=self.device)] + [self.cells[k].cell_pts for k in keys])
        if not with_fts:
            return pts
        fts = torch.vstack([torch.zeros(0, self.feature_dim, device=self.device)] + [self.cells[k].cell_features for k in keys])
        return pts, fts

    # ---- proxy points ----
    def initialize_proxy_points(self, n_proxy_points=None, default_proba_value=0.5):
        n = self.n_proxy_points if n_proxy_points is None else n_proxy_points
        dev = self.device
        self.proxy_points = self.x_min.to(dev) + (self.x_max - self.x_min).to(dev) * torch.rand(n, 3, device=dev)
        self.proxy_proba = torch.zeros(n, 1, device=dev) + default_proba_value
        self.proxy_supervision_occ = torch.ones(n, 1, device=dev)
        self.view_states = torch.zeros(n, self.n_view_state_cameras, device=dev)
        self.out_of_field = torch.ones(n, 1, device=dev)
        self.proxy_n_inside_fov = torch.zeros(n, 1, device=dev)
        self.proxy_n_behind_depth = torch.zeros(n, 1, device=dev)

    def get_proxy_indices_from_mask(self, proxy_mask):
        return torch.arange(0, self.n_proxy_points, device=self.device).view(-1, 1)[proxy_mask]

    def get_proxy_mask_from_indices(self, proxy_indices):
        mask = torch.zeros(self.n_proxy_points, device=self.device).bool()
        mask[proxy_indices.view(-1).long()] = True
        return mask

    # ---- proxy-point state updates of one MACARONS step (macarons_utils.py:2817-2912) ----
    def update_proxy_view_states(self, camera, proxy_mask, signed_distances=None, distance_to_surface=None, X_cam=None):
        """:2817-2877: OR the bin of the direction towards the camera into the view state of the masked points (only those
        whose signed distance is below `distance_to_surface`, default 3 x the proxy spacing, when distances are given): the
        reference's `+=` followed by torch.heaviside(., 0) as ONE in-place launch on the state table."""
        from .. import ops
        update_mask = proxy_mask
        if signed_distances is not None:
            if distance_to_surface is None:
                distance_to_surface = 3 * self.distance_between_proxy_points
            update_mask = torch.zeros_like(proxy_mask).bool()
            update_mask[proxy_mask] = signed_distances.view(-1) < distance_to_surface
        if X_cam is None:
            X_cam = camera.X_cam
        rows = torch.nonzero(update_mask.view(-1)).view(-1)
        ops.view_state_update_(self.view_states, rows, self.proxy_points[rows].contiguous(), X_cam.reshape(-1, 3).contiguous(),
                               self.view_state_n_elev, self.view_state_n_azim)

    def update_proxy_out_of_field(self, fov_proxy_mask):
        self.out_of_field[fov_proxy_mask] = 0.                                                       # :2879-2886

    def update_proxy_supervision_occ(self, proxy_mask, signed_distances, tol=0.):
        self.proxy_n_inside_fov[proxy_mask] += 1                                                     # :2908-2912
        self.proxy_n_behind_depth[proxy_mask] += (signed_distances.view(-1, 1) >= -tol).float()
        self.proxy_supervision_occ[proxy_mask] = ((self.proxy_n_behind_depth[proxy_mask] / self.proxy_n_inside_fov[proxy_mask])
                                                  >= self.score_threshold).float()

    def update_from_depth(self, fov_proxy_mask, camera_record, X_cam, depth, depth_mask, fill, tol=0., distance_to_surface=None,
                          return_signed_distances=False):
        """The four calls above + Camera.get_signed_distance_to_depth_maps (testers/scene.py:402-418) as one fused pass over the
        proxy points (ops.proxy_scene_update_): nothing is compacted, nothing returns to the host."""
        from .. import ops
        if distance_to_surface is None:
            distance_to_surface = 3 * self.distance_between_proxy_points
        return ops.proxy_scene_update_(self.proxy_points, fov_proxy_mask, camera_record, depth, depth_mask, fill, X_cam,
                                       distance_to_surface, tol, self.score_threshold, self.view_state_n_elev, self.view_state_n_azim,
                                       self.view_states, self.proxy_n_inside_fov, self.proxy_n_behind_depth, self.proxy_supervision_occ,
                                       self.out_of_field, return_sgn=return_signed_distances)

    def set_all_features_to_value(self, value):
        """:2931-2941.  One fill for the whole scene (the flat store's feature table is replaced): upstream's loop is two launches per
        non-empty cell -- 46 launches of 2 us each, host-bound, for the 23 surface cells of the bench scene."""
        if self.feature_dim <= 0:
            return
        st = self.flat_store()
        if st.fts.shape[0]:
            st.fts = torch.full_like(st.fts, float(value))

    # ---- coverage metrics (macarons_utils.py:2987-3056): one segmented fp64 nearest-distance launch over all cells ----
    def _csr(self, clouds):
        off = torch.zeros(len(clouds) + 1, dtype=torch.int64)
        off[1:] = torch.cumsum(torch.tensor([c.shape[0] for c in clouds], dtype=torch.int64), 0)
        pts = torch.vstack([torch.zeros(0, 3, device=self.device)] + list(clouds)).contiguous()
        return pts, off.to(self.device)

    def scene_coverage(self, recovered_scene, surface_epsilon=None):
        """(:3031-3056) fraction of this scene's points that have a point of `recovered_scene` (same grid, same cell) strictly
        within epsilon (fp64), and the number of points.  -> (coverage tensor, n_gt_pts)."""
        epsilon = 2. * self.cell_resolution if surface_epsilon is None else surface_epsilon
        keys = [k for k, c in self.cells.items() if len(c.cell_pts) > 0]
        n_gt = sum(len(self.cells[k].cell_pts) for k in keys)
        both = [k for k in keys if len(recovered_scene.cells[k].cell_pts) > 0]
        if not both:
            return torch.zeros((), dtype=torch.float64, device=self.device) / max(n_gt, 1), n_gt
        A, a_off = self._csr([self.cells[k].cell_pts for k in both])
        B, b_off = self._csr([recovered_scene.cells[k].cell_pts for k in both])
        covered = mu.covered_mask(A, B, epsilon, a_off, b_off)
        # a TENSOR divisor: dividing a device tensor by a Python scalar multiplies by its reciprocal (1 ulp off the quotient upstream's
        # CPU division returns)
        return covered.sum().double() / torch.full((), float(n_gt), dtype=torch.float64, device=self.device), n_gt

    def camera_coverage_gain(self, part_pc, surface_epsilon=None, surface_epsilon_factor=None):
        """(:2987-3029) number of not-yet-covered surface points (cell feature 0) of the cells the partial cloud touches that
        have a point of the WHOLE in-box partial cloud within epsilon (distance rounded to fp32 before the compare, as
        upstream's `.float()`)."""
        epsilon = self.cell_resolution if surface_epsilon is None else surface_epsilon
        if surface_epsilon_factor is not None:
            epsilon = epsilon * surface_epsilon_factor
        inside = self.get_pts_in_bounding_box(part_pc, return_mask=False)
        cells = [self.cells[_key(c)] for c in self.get_englobing_cells(inside, list=True)]
        cells = [c for c in cells if len(c.cell_pts) > 0]
        if not cells or len(inside) == 0:
            return 0.
        A, _ = self._csr([c.cell_pts for c in cells])
        seen = torch.cat([c.cell_features.view(-1) for c in cells])
        covered = mu.covered_mask(A, inside.contiguous(), epsilon, fp32_compare=True)
        return (covered.float() * (1. - seen)).sum()
