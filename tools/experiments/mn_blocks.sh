set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
for hb in 64 128 256 512 1024; do
  echo "GENMI_MN_HIST_BLOCKS=$hb"
  GENMI_MN_HIST_BLOCKS=$hb bash $R/tools/experiments/prof_kind.sh multinomial_tiled r03t_hb$hb 2>&1 | grep "us_per_step\|k_mn" | cut -c1-160
done
