#!/bin/bash
# builds tools/bin/conv_bench from the same kernel sources as libmrcnn_hip.so
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I caesar-mrcnn_amd/csrc \
  tools/conv_bench.hip caesar-mrcnn_amd/csrc/conv_fwd.hip caesar-mrcnn_amd/csrc/conv_wgrad.hip -o tools/bin/conv_bench
