#!/bin/bash
# instrumentation build of the back-to-back 1x1 kernel next to the product library, then scripts/b2b_trace.py per form
set -e
cd "$(dirname "$0")/../infur_amd/csrc"
make -s -j8
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DB2B_TRACE -c conv1x1_b2b.hip -o build/conv1x1_b2b_trace.o
OBJS=$(ls build/*.o | grep -v conv1x1_b2b)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libinfur_hip_trace.so $OBJS build/conv1x1_b2b_trace.o -L/opt/rocm/lib -ldl -lpthread
cd ../..
for f in ${FORMS:-1 2}; do INFUR_B2B_FORM=$f python scripts/b2b_trace.py; done
