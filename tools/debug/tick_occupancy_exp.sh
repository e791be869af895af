for cfg in "0 0" "0 12000" "15 0" "15 12000" "240 0" "240 12000" "16 0" "224 0" "31 0"; do set -- $cfg
echo -n "pad $2: "; DROPS=$1 BEATRICE_HIP_TICK_PAD_LDS=$2 bash tools/debug/tick_drop.sh; done
