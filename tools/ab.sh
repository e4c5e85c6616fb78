#!/bin/bash
# Same-box A/B of tools/stage_times.py under different environments, alternating, ROUNDS times.
# usage: tools/ab.sh "TAG1:VAR=val VAR2=val" "TAG2:..." ...   (a variant may be just "TAG:")
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    tag=${v%%:*}; envs=${v#*:}
    env $envs QB_TAG=$tag python tools/stage_times.py || exit 1
  done
done
