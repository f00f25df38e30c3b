"""removePoints / makeBVFeature -- drop-ins for reference src/data_process/kitti_bev_utils.py:18-76 that rasterise on
the HIP device (cy_bev_rasterize) instead of numpy's lexsort + unique.

The reference filters in removePoints and rasterises in makeBVFeature.  makeBVFeature keeps that contract: its input is
taken as ALREADY filtered and z-shifted (what removePoints returns), whatever its container type.  The device kernel
can also do the filter itself in the same pass: ``makeBVFeature(points, ..., raw=True)`` takes unfiltered points.
Return type follows the input: numpy in -> numpy float64 [3, H, W] (as the reference), device tensor in -> device
float32 tensor, which is what the training loop wants (kitti_dataset.py:115 casts to float32 anyway)."""
import numpy as np
import torch

from .. import ops
from ..config import kitti_config as cnf

_WS = {}


def removePoints(PointCloud, BoundaryCond):
    """Points inside the closed box of BoundaryCond, z shifted so that minZ maps to 0 (reference :18-34)."""
    if torch.is_tensor(PointCloud):
        p = PointCloud
        m = ((p[:, 0] >= BoundaryCond['minX']) & (p[:, 0] <= BoundaryCond['maxX']) & (p[:, 1] >= BoundaryCond['minY']) &
             (p[:, 1] <= BoundaryCond['maxY']) & (p[:, 2] >= BoundaryCond['minZ']) & (p[:, 2] <= BoundaryCond['maxZ']))
        out = p[m].clone()
        out[:, 2] -= BoundaryCond['minZ']
        return out
    p = np.asarray(PointCloud)
    m = ((p[:, 0] >= BoundaryCond['minX']) & (p[:, 0] <= BoundaryCond['maxX']) & (p[:, 1] >= BoundaryCond['minY']) &
         (p[:, 1] <= BoundaryCond['maxY']) & (p[:, 2] >= BoundaryCond['minZ']) & (p[:, 2] <= BoundaryCond['maxZ']))
    out = p[m].copy()
    out[:, 2] = out[:, 2] - BoundaryCond['minZ']
    return out


def makeBVFeature(PointCloud_, Discretization, bc, height=None, width=None, raw=False):
    """[n,4] points (x, y, z, intensity) -> [3, H, W] maps (intensity, height, density), reference :37-76.
    As in the reference the points are those removePoints returned (inside the box, z - minZ).  ``raw=True``: the
    points are unfiltered LiDAR points and the kernel applies removePoints' box filter and z shift itself."""
    H = cnf.BEV_HEIGHT if height is None else height
    W = cnf.BEV_WIDTH if width is None else width
    as_numpy = not torch.is_tensor(PointCloud_)
    pts = torch.as_tensor(np.ascontiguousarray(PointCloud_, dtype=np.float32)) if as_numpy else PointCloud_.float()
    pts = pts.to('cuda') if not pts.is_cuda else pts
    bounds = [bc['minX'], bc['maxX'], bc['minY'], bc['maxY'], bc['minZ'], bc['maxZ']]
    zshift, max_height = bc['minZ'], float(abs(bc['maxZ'] - bc['minZ']))
    if not raw:   # already filtered and z already relative to minZ: no z filter, no shift
        bounds[4], bounds[5], zshift = -3.0e38, 3.0e38, 0.0
    key = (H, W, str(pts.device))
    if key not in _WS:
        _WS[key] = ops.bev_workspace(H, W, pts.device)
    out = ops.bev_rasterize(pts, bounds, Discretization, H, W, _WS[key], zshift=zshift, max_height=max_height)
    return out.double().cpu().numpy() if as_numpy else out
