"""bench.py with another build of the library (HVR_BENCH_LIB=path): A/B runs of two builds on one box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hvrnet_amd import native
if os.environ.get('HVR_BENCH_LIB'):
    native.LIB_PATH = os.path.abspath(os.environ['HVR_BENCH_LIB'])
import bench
bench.main(sys.argv[1:])
