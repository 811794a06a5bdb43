#!/bin/bash
# Static evidence for the hand-written encoder kernels: register / scratch / LDS use and the instruction mix of each
# kernel's code, from the gfx950 assembly hipcc emits (no GPU needed).  Usage: scripts/isa_report.sh > profiles/<name>.txt
set -e
cd "$(dirname "$0")/../leann_amd/csrc"
TMP=$(mktemp -d)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -S --cuda-device-only"
report() {  # file extra-flags kernel-regex
    local f=$1 extra=$2 pat=$3
    /opt/rocm/bin/hipcc $FLAGS $extra $f -o $TMP/k.s 2>/dev/null
    for sym in $(grep -oE "^_ZN2lm[A-Za-z0-9_]*:" $TMP/k.s | tr -d ':' | grep -E "$pat"); do
        a=$(grep -n "^${sym}:" $TMP/k.s | head -1 | cut -d: -f1); b=$(grep -n "amdhsa_kernel ${sym}\$" $TMP/k.s | head -1 | cut -d: -f1)
        [ -z "$a" ] || [ -z "$b" ] && continue
        sed -n "${a},${b}p" $TMP/k.s > $TMP/kk.s
        vg=$(grep -E "\.set ${sym}\.num_vgpr," $TMP/k.s | sed 's/.*, //'); ag=$(grep -E "\.set ${sym}\.num_agpr," $TMP/k.s | sed 's/.*, //')
        sc=$(grep -E "\.set ${sym}\.private_seg_size," $TMP/k.s | sed 's/.*, //')
        printf "%s\n  file %s  vgpr=%s agpr=%s scratch_bytes=%s instructions=%s\n  " "$(echo $sym | c++filt)" "$f" "$vg" "$ag" "$sc" "$(grep -cE '^\s+[vsdgb][_a-z]' $TMP/kk.s)"
        for p in v_mfma v_exp_f32 v_rcp_f32 v_pk_ v_cndmask v_cmp v_cvt v_accvgpr v_readlane v_writelane ds_read ds_write global_load global_store s_waitcnt s_nop s_barrier scratch_; do
            printf "%s=%s " $p $(grep -c "$p" $TMP/kk.s)
        done
        printf "\n"
    done
}
echo "# gfx950 static report, $(/opt/rocm/bin/hipcc --version | grep -m1 -i 'hip version')"
report lm_encoder_ops.hip "" "k_attn_varlen_hd32ILi8|k_add_layernorm_f16ILi1"
report lm_attn_v2.hip "-mllvm -amdgpu-mfma-vgpr-form=1" "k_attn_varlen_hd32_v2ILi8"
report lm_encoder_ops2.hip "" "k_add_layernorm_f16_r16ILi3ELb1ELb1|k_embed_layernorm_f16ILi3ELb1|k_meanpool"
report lm_mlp_fused.hip "" "k_mlp_fused_h384"
report lm_linear_h384.hip "" "k_linear_h384"
rm -rf $TMP
