#!/bin/bash
# One GPU call of the kernel work loop: (1) traversal parity tests on the library in the tree, (2) A/B bench lines of the
# builds under variants/ and of the tree's library. Usage: tools/r3_step.sh <tag> [pytest -k expression]
set -u
tag=${1:-x}
sel=${2:-"closest_hit or shadow_query or traversal or device_built or deep_tree or stack_layouts or pass_finishing or per_sample_radiance"}
cd "$(dirname "$0")/.."
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "$sel" > gpurun_out/${tag}_tests.log 2>&1 || { tail -30 gpurun_out/${tag}_tests.log; exit 1; }
tail -2 gpurun_out/${tag}_tests.log
cp mitsuba_amd/libmtsamd.so variants/z_tree.so
bash tools/variants.sh $tag
rm -f variants/z_tree.so
