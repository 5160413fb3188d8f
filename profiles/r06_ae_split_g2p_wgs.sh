for w in 0 1536 2304 4394; do
  for ov in 1; do
    if [ $w = 0 ]; then unset MPMHIP_G2P_WGS; else export MPMHIP_G2P_WGS=$w; fi
    MPMHIP_TILE_OVERLAP=$ov python bench.py --virtual 8 --steps 40 --warmup 10 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('g2p_wgs $w overlap $ov per_rank_ms', d['per_rank_ms_serial_no_events'], d['rank0_phases_ms'])"
  done
done
