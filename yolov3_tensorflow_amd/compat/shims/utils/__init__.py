# coding: utf-8
"""The reference's `utils` package name over yolov3_tensorflow_amd.utils (compat layer; see
yolov3_tensorflow_amd/compat/__init__.py)."""
