# the round's second profile set: final_scene (tag r2_e) and the 10 M-sphere scene (tag r2_e_big)
bash tools/gpu_r2_profile.sh r2_e || exit 1
bash tools/gpu_big_profile.sh 1581 64 r2_e_big || exit 1
