#!/bin/bash
# Run HERE after `gpurun -- 'bash tools/collect_profiles.sh'` (and tools/collect_final.sh): copies the merged summaries from
# gpurun_out/profiles_new/ and gpurun_out/final/ into profiles/ under the round prefix given as $1 (e.g. r04).
set -e
r=${1:?round prefix, e.g. r04}
cd "$(dirname "$0")/.."
for f in gpurun_out/profiles_new/bench_*_kernel_stats.csv gpurun_out/profiles_new/bench_*_under_rocprof.log gpurun_out/profiles_new/pmc_*.json; do
  [ -f "$f" ] || continue
  b=$(basename "$f")
  case "$b" in pmc_*_[0-9].log) continue;; esac
  cp "$f" "profiles/${r}_$b"
done
if [ -d gpurun_out/final ]; then
  for f in gpurun_out/final/*; do
    case "$f" in *.err) continue;; esac
    [ -s "$f" ] && cp "$f" "profiles/${r}_$(basename "$f")"
  done
fi
ls profiles | grep "^${r}_" | wc -l
