"""Darknet cfg lowering (graph), executor (engine) and the reference-compatible model classes."""
