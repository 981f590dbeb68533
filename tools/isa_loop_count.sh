#!/bin/bash
# Instruction count of a kernel's main loop, per basic block -- the metric the two blend loops are written against
# (a wave on the critical path retires one instruction per ~6.6 cycles whatever its kind).
#   bash tools/isa_loop_count.sh forward  [records_per_iteration=6]     k_render      (gsr_forward.hip, -ffp-contract=off)
#   bash tools/isa_loop_count.sh backward [records_per_iteration=4]     k_render_bwd  (gsr_backward.hip, -ffp-contract=fast)
#   ... forward_fast / backward_fast: the GsrSettings.fast_blend instances of the two kernels
# Prints the static totals of the first depth-1 loop and one line per basic block (n = instructions, v = VALU, s = SALU,
# m = scalar memory, then the block's branches); the hot path is read off the branch structure by hand.
set -eu
which=${1:-forward}
here=$(cd "$(dirname "$0")/../gaussianavatars_amd/csrc" && pwd)
case "$which" in
  forward)       src=gsr_forward.hip;  fp=off;  kern=_ZN3gsr8k_renderILb0ELi0ELb0EE;            r=${2:-6};;
  forward_fast)  src=gsr_forward.hip;  fp=off;  kern=_ZN3gsr8k_renderILb1ELi0ELb0EE;            r=${2:-6};;   # GsrSettings.fast_blend
  backward)      src=gsr_backward.hip; fp=fast; kern=_ZN3gsr12k_render_bwdILb0ELb0EE;   r=${2:-4};;
  backward_fast) src=gsr_backward.hip; fp=fast; kern=_ZN3gsr12k_render_bwdILb0ELb1EE;   r=${2:-4};;
  *) echo "forward | forward_fast | backward | backward_fast"; exit 2;;
esac
tmp=$(mktemp -d)
(cd "$here" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=$fp -fno-slp-vectorize ${ISA_DEFS:-} -S --cuda-device-only $src -o $tmp/k.s 2>/dev/null)
awk -v k="$kern" '$0 ~ "^"k".*:" {on=1} on && /s_endpgm/ {on=0} on {print}' $tmp/k.s > $tmp/kernel.s
hdr=$(grep -m1 "Loop Header: Depth=1" $tmp/kernel.s | sed 's/:.*//; s/^\.L//')
awk -v h="$hdr" -v r="$r" '
function flush() { if (lbl != "") printf "%-10s n=%3d v=%3d s=%3d m=%2d  %s\n", lbl, n, v, s, m, br }
/^\.LBB/ { flush(); inloop = (index($0, "Header=" h " ") || $0 ~ ("^\\.L" h ":")); lbl = inloop ? $1 : ""; n=v=s=m=0; br=""; next }
/^[ \t]*;/ {next}
/^[ \t]*\./ {next}
lbl != "" && NF>0 { n++; N++; if ($1 ~ /^v_/) {v++; V++} else if ($1 ~ /^s_load/) {m++; M++} else {s++; S++}; if ($1 ~ /branch/) br = br " " $1 ">" $2 }
END { flush(); printf "loop %s static: %d instructions (%.1f per record if every block ran once): valu %d salu %d smem %d\n", h, N, N/r, V, S, M }
' $tmp/kernel.s
grep -E "; (NumVgprs|ScratchSize|Occupancy)" $tmp/k.s | head -0
rm -rf $tmp
