"""BEV geometry of the KITTI front view -- the numbers of reference src/config/kitti_config.py:13-36 that the
rasteriser reads, exposed under the names the reference's call sites use (``cnf.boundary``, ``cnf.DISCRETIZATION``...).

The map covers 50 m ahead of the car and 25 m to either side at 608 x 608 pixels; points are kept between 2.73 m
below and 1.27 m above the sensor."""

_AHEAD_M, _HALF_WIDTH_M = 50, 25
_Z_RANGE_M = (-2.73, 1.27)


def _box(x0, x1):
    return dict(minX=x0, maxX=x1, minY=-_HALF_WIDTH_M, maxY=_HALF_WIDTH_M, minZ=_Z_RANGE_M[0], maxZ=_Z_RANGE_M[1])


boundary = _box(0, _AHEAD_M)            # in front of the vehicle
boundary_back = _box(-_AHEAD_M, 0)      # behind it

BEV_HEIGHT = 608                        # pixels along x (0 .. 50 m)
BEV_WIDTH = 608                         # pixels along y (-25 .. 25 m)
DISCRETIZATION = (boundary['maxX'] - boundary['minX']) / BEV_HEIGHT   # metres per pixel
