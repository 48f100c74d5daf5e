#!/bin/bash
# Developer tool: the CPU-side code under AddressSanitizer + UndefinedBehaviorSanitizer (GPU ASan and
# XNACK runs are not available on this pool).  Builds sanitized copies of libsortham_host.so (host
# optimizer, PathHandler, TrajectoryVisualizer, the compiled tick loop) and of the oracle under
# /tmp/smpc_asan and runs the CPU tests that exercise them with the sanitizer runtimes preloaded.
#   tools/sanitize_cpu.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/smpc_asan
mkdir -p $OUT
make -s -C $ROOT/mpcholonavigation_amd/csrc
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
( cd $ROOT/mpcholonavigation_amd/csrc && g++ -std=c++17 -fPIC -shared $SAN -o $OUT/libsortham_host.so \
    ../host/optimizer.cpp ../host/optimizer_c.cpp ../host/path_handler.cpp ../host/path_handler_c.cpp ../host/tick_loop.cpp \
    -L.. -lsmpc -Wl,-rpath,$ROOT/mpcholonavigation_amd )
( cd $ROOT/oracle && g++ -std=c++17 -shared -fPIC -I../include -ffp-contract=off $SAN -o $OUT/liboracle.so smpc_oracle.cpp smpc_oracle_host.cpp )
cat > $OUT/run.py <<PY
import ctypes as C, sys
sys.path.insert(0, "$ROOT")
import pytest
import mpcholonavigation_amd.host_optimizer as H
import oracle.loader as L
from mpcholonavigation_amd import _abi as A
H.LIB_PATH = "$OUT/libsortham_host.so"
lib = C.CDLL("$OUT/liboracle.so"); A.bind(lib, L._PROTOTYPES); L._libs["liboracle.so"] = lib
sys.exit(pytest.main(["-q", "-m", "not gpu", "-p", "no:cacheprovider", "$ROOT/tests/test_path_handler_cpu.py", "$ROOT/tests/test_host_cpu.py",
                      "$ROOT/tests/test_oracle_reference_kats.py", "$ROOT/tests/test_golden.py"]))
PY
cd $ROOT
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python3 $OUT/run.py
