"""Scan-sequence recipes that configure and call the hot path."""
from dosma_amd.scan_sequences.qdess import QDess  # noqa: F401
from dosma_amd.scan_sequences.recipes import Cones, CubeQuant, Mapss  # noqa: F401
