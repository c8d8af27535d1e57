set -e
cd $GRAFT_REPO_ROOT
V=pathtracer_amd/lib/variants
for rep in 1 2; do
for v in "" top32 top64 top128 top0s9; do
  if [ -z "$v" ]; then timeout -k 10 120 python3 tools/time_closest.py 16777216 10; else PTX_AMD_LIB=$PWD/$V/libptx_$v.so timeout -k 10 120 python3 tools/time_closest.py 16777216 10; fi
done; done
