#!/bin/bash
# The 13 NIfTI volumes the reference ships (/root/reference/dataset: one ceT1 and three hrT2 cases with labels, five
# translated "fake" volumes) -> tests/golden/_refdata/ (git-ignored: 23 MB of the reference's sample data are not committed;
# they travel to the GPU box with the gpurun snapshot like the built .so).  tools/dice_real.py reads them from there.
set -e
src=${1:-/root/reference/dataset}
dst="$(cd "$(dirname "$0")/.." && pwd)/tests/golden/_refdata"
mkdir -p "$dst"
cp -r "$src"/. "$dst"/
find "$dst" -type f ! -name "*.nii.gz" -delete
find "$dst" -name "*.nii.gz" | wc -l
