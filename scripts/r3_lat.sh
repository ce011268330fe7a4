set -e
R=$GRAFT_REPO_ROOT; cd $R
for per in 4 8 16; do
echo "NG_VEC_PER=$per"
NG_VEC_PER=$per timeout -k 10 300 python scripts/latbench.py 2>/dev/null | tail -1
done
