"""bench.py with a privately built library (tuning A/B only):  HVR_BENCH_LIB=dbg/libhvr_x.so python tools/probe/bench_lib.py [bench args]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hvrnet_amd import native
if os.environ.get('HVR_BENCH_LIB'):
    native.LIB_PATH = os.path.abspath(os.environ['HVR_BENCH_LIB'])
import bench
bench.main(sys.argv[1:])
