for S in 10240 10000; do
for cfg in "0 0" "1 0" "0 32" "1 32" "1 16" "0 0"; do set -- $cfg; 
  echo "S=$S XY=$1 LPB=$2"; 
  if [ "$2" = "0" ]; then TRK_CF_XY=$1 python tools/v2_mode_probe.py --samples $S --rounds 2 1 2>&1 | tail -1;
  else TRK_CF_XY=$1 TRK_CF_LPB=$2 python tools/v2_mode_probe.py --samples $S --rounds 2 1 2>&1 | tail -1; fi
done; done
