for D in "" "-DPS_ABL_NOREAD" "-DPS_ABL_NOMFMA" "-DPS_ABL_NODMA" "-DPS_NO_ILV" "-DPS_ABL_NOREAD -DPS_ABL_NODMA"; do
  echo "#### DEFS: $D"
  for c in "quar_256_256_3x3d2 0x10240" "8th_256_256 0x10120" "half_64_64_3x3 0x10140"; do set -- $c; TRACE_DEFS="$D" TRACE_CFG=$2 python tools/trace_ps.py $1 2>&1 | grep -v amdgpu.ids; done
done
