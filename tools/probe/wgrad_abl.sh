# timing probes of the record weight-gradient kernel on the dominant shape (inside gpurun): bash tools/probe/wgrad_abl.sh "0 1 2 3 4 5"
for a in ${1:-0 1 2 3 4}; do echo "ABL $a"; D2T_WGRAD_ABL=$a bash tools/profile_train.sh wgabl$a 32 2 | grep "'2048', '9', '7'"; done
