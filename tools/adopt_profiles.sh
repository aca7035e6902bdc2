#!/bin/bash
# Run HERE after `gpurun -- 'bash tools/collect_profiles.sh'` and / or `gpurun -- 'bash tools/collect_final.sh'`: copies the
# summaries those runs produced (their MANIFEST: the local gpurun_out/ may still hold files of earlier rounds) into profiles/
# under the round prefix given as $1 (e.g. r04).
set -e
r=${1:?round prefix, e.g. r04}
cd "$(dirname "$0")/.."
for d in gpurun_out/profiles_new gpurun_out/final; do
  [ -f "$d/MANIFEST" ] || { echo "no $d/MANIFEST: nothing adopted from $d"; continue; }
  while read -r b; do
    [ "$b" = MANIFEST ] && continue
    [ -s "$d/$b" ] && cp "$d/$b" "profiles/${r}_$b"
  done < "$d/MANIFEST"
done
ls profiles | grep -c "^${r}_"
