#!/bin/bash
# Writes the LAB form of the antenna-factored fringe kernels to tools/bin/lab_src/fringe_mfma.hip (git-ignored) and prints its
# path: the product source bayeslim_amd/csrc/fringe_mfma.hip + fringe_mfma_lab.patch.  `--check` only tests that the patch
# still applies (CPU test tests/test_host_logic.py::test_lab_patch_applies_to_the_product_kernels).
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
if [ "$1" = "--check" ]; then
    exec patch --dry-run -s -p1 -d $root -i $root/tools/lab/fringe_mfma_lab.patch
fi
out=$root/tools/bin/lab_src
mkdir -p $out
cp $root/bayeslim_amd/csrc/fringe_mfma.hip $out/fringe_mfma.hip
patch -s -p3 -d $out -i $root/tools/lab/fringe_mfma_lab.patch
echo $out/fringe_mfma.hip
