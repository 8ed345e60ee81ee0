# knock-out / variant rows of tools/wino44_ko.py inside ONE call: r04_ko.sh <lib> [<lib> ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ko; mkdir -p $O; cd $R
echo "library                      208:32->64  104:64->128  52:128->256  26:256->512  13:512->1024   (us per launch, bs=32)" | tee $O/ko.txt
for L in "$@"; do Y3_LIB_PATH=$L timeout 60 python tools/wino44_ko.py 2>&1 | tail -1 | tee -a $O/ko.txt; done
