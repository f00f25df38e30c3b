"""Darknet cfg files (cfg/) and the KITTI BEV geometry constants."""
