# memory-skeleton probes of the row-gather assembly
set -e
cd $GRAFT_REPO_ROOT
for m in 64 65 67 71 70 0 1 2 4 5 6 8 9 10; do timeout -k 10 60 scripts/gatherbench.bin $m; done
timeout -k 10 60 scripts/gatherbench.bin 7 8 8
timeout -k 10 60 scripts/gatherbench.bin 7 32 8
