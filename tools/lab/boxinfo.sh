#!/usr/bin/env bash
# What kind of box is this?  Identity, clocks, partition modes, power cap of the visible GPU(s) -- next to the arena levels of the same call.
echo "== $(date -u +%FT%TZ) host $(hostname)"
(rocm-smi --showuniqueid --showserial --showvbios --showmemorypartition --showcomputepartition --showpower --showmaxpower --showtemp --showclocks --showperflevel --showmemvendor 2>&1 | grep -v "^=\|^$" | head -60) || true
for f in /sys/class/drm/card*/device/{pp_dpm_mclk,pp_dpm_fclk,pp_dpm_sclk,pp_dpm_socclk,current_memory_partition,current_compute_partition,mem_info_vram_vendor,mem_info_vram_total,power_dpm_force_performance_level}; do
    [ -r "$f" ] && echo "$f: $(tr '\n' ' ' < "$f")"
done 2>/dev/null | head -60
