#!/bin/bash
# usage: scripts/quick_bench.sh tag1 tag2 ...   ("base" = default library)
for tag in "$@"; do
  if [ "$tag" = "base" ]; then export ANM_BUILD_TAG=; else export ANM_BUILD_TAG=$tag; fi
  for mi in 100 20; do
    r=$(python bench.py --steps 100 --warmup 10 --no-cpu-baseline --max-iter $mi 2>&1 | grep -o "kernel_ms\": [0-9.]*")
    echo "$tag max_iter=$mi $r"
  done
done
