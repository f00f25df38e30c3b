"""LiDAR -> BEV rasterisation on the device (kitti_bev_utils)."""
