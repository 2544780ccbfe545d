for v in "" "CSMRI_GCONV_NOSPLIT_TILES=600 CSMRI_GCONV_BLOCKS=512" "CSMRI_GCONV_NOSPLIT_TILES=600" "CSMRI_GCONV_NOSPLIT_TILES=600 CSMRI_GCONV_BLOCKS=1024" "CSMRI_GLDS_STAGES=4" "CSMRI_GLDS_STAGES=1"; do
  echo "== $v"
  env $v python tools/bench_conv.py vgg4_2 vgg3_2 vgg5_2 u128 disc3 dgrad 2>&1 | grep -v amdgpu.ids
done
