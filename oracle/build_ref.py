"""TEST INFRASTRUCTURE: compile the reference's own CPU NMS, unmodified, from where it lies.

    python oracle/build_ref.py      ->  oracle/_ref/hvr_ref_nms_cpu.so

Source: /root/reference/mmdet/ops/nms/src/nms_cpu.cpp (a single-file torch extension; torch is
part of this image, so nothing is stubbed).  The output is git-ignored but travels to the GPU
box with the snapshot; /root/reference itself never does, so this is a no-op there.
It compiles as-is against torch 2.10 (deprecation warnings only).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = '/root/reference/mmdet/ops/nms/src/nms_cpu.cpp'
OUT_DIR = os.path.join(HERE, '_ref')
NAME = 'hvr_ref_nms_cpu'


def build():
    if not os.path.exists(SRC):
        return None
    os.makedirs(OUT_DIR, exist_ok=True)
    out = os.path.join(OUT_DIR, NAME + '.so')
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(SRC):
        return out
    from torch.utils.cpp_extension import load
    os.environ.setdefault('MAX_JOBS', '4')
    load(name=NAME, sources=[SRC], build_directory=OUT_DIR, extra_cflags=['-O2'], verbose=False)
    return out


def load_ref():
    """Import the built module (None when it was never built)."""
    out = os.path.join(OUT_DIR, NAME + '.so')
    if not os.path.exists(out):
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location(NAME, out)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == '__main__':
    print(build() or 'reference tree not present; nothing built')
