#!/bin/bash
# Fails the build when a function that MUST stay out of line got inlined.
#   wave_ticket  (csrc/wave_ops.h)    — inlined into a loop with a wave-uniform `continue`, the lane-0 atomic is threaded
#                                       through the back edge and the wave splits (profiles/r01_notes.md #21)
#   sort_records (csrc/region_sort.h) — the lane-0 introsort over LDS; see the note at its definition
# For every object whose SOURCE names the function, the gfx950 code object must define it as a FUNC symbol and contain
# at least one s_swappc_b64 (a real call).  usage: check_outofline.sh <build dir> <src dir>
set -e
OBJDUMP=/opt/rocm/lib/llvm/bin/llvm-objdump
build=$1; src=$2; tmp=$(mktemp -d); trap 'rm -rf "$tmp"' EXIT
rc=0
for s in "$src"/*.hip; do
    f=$(basename "$s" .hip)
    for fn in wave_ticket sort_records; do
        grep -q "\b$fn(" "$s" || continue
        cp "$build/$f.o" "$tmp/" && (cd "$tmp" && $OBJDUMP --offloading "$f.o" >/dev/null 2>&1)
        co=$(ls "$tmp"/$f.o.*gfx950 2>/dev/null | head -1)
        [ -n "$co" ] || { echo "check_outofline: no gfx950 code object in $f.o"; rc=1; continue; }
        nsym=$($OBJDUMP -t "$co" | grep -E ' F \.text' | grep -c "$fn" || true)
        ncall=$($OBJDUMP -d "$co" | grep -c s_swappc_b64 || true)
        if [ "$nsym" -lt 1 ] || [ "$ncall" -lt 1 ]; then
            echo "check_outofline: $fn was INLINED in $f.hip (symbols $nsym, calls $ncall)"; rc=1
        fi
    done
done
[ $rc -eq 0 ] && echo "check_outofline: ok"
exit $rc
