#!/bin/bash
# tools/kernel_resources.sh FILE.{o,so} [name-filter] — VGPRs / SGPRs / spills / scratch of every gfx950 kernel in a HIP object
# (code-object metadata, what the hardware allocates from).
set -e
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
B=/opt/rocm/lib/llvm/bin
TGT=$($B/clang-offload-bundler --list --type=o --input=$T/fat.bin | grep gfx950 | head -1)
$B/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=$TGT --output=$T/dev.co
$B/llvm-readelf --notes $T/dev.co | awk -v f="${2:-}" '
  /\.name:/ {name=$2}
  /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.vgpr_spill_count:/ {vs=$2} /\.sgpr_spill_count:/ {ss=$2}
  /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2}
  /\.wavefront_size:/ { if (name ~ f) printf "%-110s vgpr %3s sgpr %3s vspill %3s sspill %3s scratch %4s\n", name, v, s, vs, ss, p }'
rm -rf $T
