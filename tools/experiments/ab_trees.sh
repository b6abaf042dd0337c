#!/bin/bash
# Interleaved same-box A/B of two source trees (e.g. a `git worktree add ab_base HEAD` copy with its own built library
# against the working tree):   tools/ab_trees.sh <log> <rounds> <treeA> <treeB> -- <bench.py args...>
log=$1; rounds=$2; a=$3; b=$4; shift 5
for rep in $(seq $rounds); do for t in $a $b; do
  echo "tree=$t" >> $log
  python $t/bench.py "$@" 2>&1 | tail -1 >> $log
done; done
