#!/bin/bash
# the triangle-phase threshold of the traversal loop on the four workloads
set -u
cd "$(dirname "$0")/.."
for w in veach cornell config3 config5; do
  echo "== $w"
  steps=6; [ $w = config3 ] && steps=2; [ $w = config5 ] && steps=2; [ $w = cornell ] && steps=20
  WORKLOAD=$w STEPS=$steps bash tools/ab_env.sh "MTSAMD_TRI_SHARE=16" "MTSAMD_TRI_SHARE=24" "MTSAMD_TRI_SHARE=32" "MTSAMD_TRI_SHARE=40" "MTSAMD_TRI_SHARE=48" | cut -c1-130 || exit 1
done
