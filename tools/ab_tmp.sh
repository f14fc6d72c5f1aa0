for l in libidh_prev libidh libidh_prev libidh; do echo $l; IDH_LIB=implicit-depth_amd/lib/$l.so python tools/perf_levels.py 32 2>/dev/null | head -1; done
for l in libidh_prev libidh; do echo $l; IDH_LIB=implicit-depth_amd/lib/$l.so python tools/perf_levels.py 1 2>/dev/null | head -1; IDH_LIB=implicit-depth_amd/lib/$l.so python tools/perf_levels.py 4 2>/dev/null | head -1; done
