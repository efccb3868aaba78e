#!/bin/bash
# tools/gpu.sh [--timeout S] '<command>': gpurun with the commit being measured recorded in .head_at_push (the snapshot has
# no .git; tools/profile_bench.sh copies it into its summaries so that a profile says which code it measured).
T=900
if [ "$1" = "--timeout" ]; then T=$2; shift 2; fi
cd "$(dirname "$0")/.."
echo "$(git rev-parse --short=12 HEAD)$(git diff --quiet || echo +dirty)" > .head_at_push
exec /usr/local/graft/bin/gpurun --timeout $T -- "$@"
