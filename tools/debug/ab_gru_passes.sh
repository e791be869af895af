for cfg in "512 - 4" "1024 - 4" "1024 - 1" "1024 - 2" "2048 - 4" "512 - 1" "256 - 4"; do
  for p in 1 2; do
    echo "B,H=$cfg passes=$p: $(BEATRICE_HIP_GRU_PASSES=$p python tools/debug/time_tick.py $cfg 2>/dev/null | grep -E 'TimeTickLaunch\((64|16)\)|loop' | sed -E 's/TimeTickLaunch\(([0-9]+)\): //; s/loop without drain: /loop /; s/ per tick.*//' | tr '\n' ' ')"
  done
done
