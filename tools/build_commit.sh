#!/bin/bash
# tools/build_commit.sh <commit> <name> -- the library of an OLDER commit, built from a scratch worktree with that commit's own build script,
# as tools/bin/libsextans_<name>.so (git-ignored, travels with gpurun): same-box A/B of two HEADs (tools/nasa_ab.py, tools/ab.py).
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/sx_worktree_$2
rm -rf $W; git -C $REPO worktree prune; git -C $REPO worktree add -f --detach $W $1 > /dev/null 2>&1
(cd $W && python -m sextans_amd.build > /dev/null 2>&1)
mkdir -p $REPO/tools/bin
cp $W/sextans_amd/lib/libsextans_amd.so $REPO/tools/bin/libsextans_$2.so
git -C $REPO worktree remove --force $W
echo built tools/bin/libsextans_$2.so from $1
