"""BEV geometry constants of the KITTI front view, the part of reference src/config/kitti_config.py:13-36 the
rasteriser reads: 50 m x 50 m in front of the car at 608 x 608 pixels, heights between -2.73 m and 1.27 m."""

boundary = {"minX": 0, "maxX": 50, "minY": -25, "maxY": 25, "minZ": -2.73, "maxZ": 1.27}
boundary_back = {"minX": -50, "maxX": 0, "minY": -25, "maxY": 25, "minZ": -2.73, "maxZ": 1.27}

BEV_WIDTH = 608   # across the y axis, -25 m .. 25 m
BEV_HEIGHT = 608  # across the x axis, 0 m .. 50 m
DISCRETIZATION = (boundary["maxX"] - boundary["minX"]) / BEV_HEIGHT
