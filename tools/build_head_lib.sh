#!/bin/bash
# build 3pre_amd/lib/libpre3_head.so from the committed sources (git HEAD) for same-box A/B runs (tools/ab_head.sh)
set -e
rm -rf /tmp/headsrc && mkdir -p /tmp/headsrc
git archive HEAD 3pre_amd/csrc include | tar -x -C /tmp/headsrc
make -C /tmp/headsrc/3pre_amd/csrc >/dev/null
cp /tmp/headsrc/3pre_amd/lib/libpre3.so 3pre_amd/lib/libpre3_head.so
