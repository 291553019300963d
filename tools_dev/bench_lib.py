# developer helper: run bench.py against another build of the library (WMX_TOOL_LIB=path), e.g. an experiment variant
import os, sys, runpy
os.environ.setdefault('WMIX_AMD_ALLOW_VARIANT_BUILD', '1')  # this tool exists to load variants on purpose
sys.path.insert(0, '.')
from wmix_amd import _lib
if os.environ.get('WMX_TOOL_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['WMX_TOOL_LIB'])
sys.argv = ['bench.py'] + sys.argv[1:]
runpy.run_path('bench.py', run_name='__main__')
