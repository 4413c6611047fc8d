#!/bin/bash
# builds tools/read_lab/ring_lab against the library's object files (make -C linkteller_amd/csrc first)
set -e
cd "$(dirname "$0")/../.."
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Ilinkteller_amd/csrc -c tools/read_lab/ring_lab.hip -o /tmp/ring_lab.o
cd linkteller_amd/csrc
hipcc --offload-arch=gfx950 /tmp/ring_lab.o lt_core.o lt_gemm.o lt_spmm.o lt_forward.o lt_influence.o lt_gcn3.o lt_dp.o -o ../../tools/read_lab/ring_lab
