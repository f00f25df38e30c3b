"""MI355X-native hot path for Complex-YOLOv4.  Host-side mirror of the reference's
model / geometry / post-processing interfaces over the C-ABI library ``libcyolo_hip.so``."""
__version__ = '0.1.0'
