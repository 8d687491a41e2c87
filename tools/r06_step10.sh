cd ${GRAFT_REPO_ROOT:-/root/repo}
D=node-speex-resampler_amd/ab/libspeexhip_diag.so
for E in "SPEEXHIP_TOUCH=1" "SPEEXHIP_TOUCH=1 SPEEXHIP_PP=1" "SPEEXHIP_KS_UNSPLIT=0"; do
  echo "== $E"
  env SPEEXHIP_LIB_PATH=$D SPEEXHIP_MODE=fast $E python -m pytest -x -q -m gpu tests/test_gpu_parity.py -k "(every_golden_case or many_rates or window_layout_variants or tap_range_shares or int16_window or mono_rows or mono_packed or many_generation or eight_channel or fp64_accumulate_period or control_scripts_fast or ragged or float_entry) and not phase_pair and not tap_rows_fetched" 2>&1 | grep -v "^$" | tail -25
done
