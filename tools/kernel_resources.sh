#!/bin/bash
# Register, scratch and LDS use of every kernel in the library, as the compiler reports them
# (-Rpass-analysis=kernel-resource-usage; device code only, nothing is linked):  tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/pathtracer_amd/csrc
for f in ptx_api.hip render.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC --cuda-device-only -c -o /dev/null \
    -Rpass-analysis=kernel-resource-usage $C/$f 2>&1 | sed -e 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
    awk '/remark: Function Name:/ {name=$NF} /remark: +VGPRs:/ {v=$NF} /remark: +AGPRs:/ {a=$NF} /ScratchSize/ {s=$NF} /Occupancy/ {o=$NF} /VGPRs Spill/ {sp=$NF} /LDS Size/ {print name, "vgprs", v, "agprs", a, "scratch_bytes_per_lane", s, "vgpr_spills", sp, "waves_per_simd", o, "lds_bytes", $NF}'
done | while read name rest; do echo "$(echo $name | c++filt | sed -E 's/\(ptx::.*$//; s/\(unsigned.*$//; s/\(float.*$//') | $rest"; done | sort -u
