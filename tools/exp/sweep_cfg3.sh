cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python tools/exp/cfg_stages.py 3 2>&1 | grep -E "xk_|total" | head -3; }
run XK_NOP=1
run XK_CAQR_ARITY1=20
run XK_CAQR_CHALF=16
run XK_CAQR_CUS=512
run XK_CAQR_CUS=128
run XK_CAQR_ADAPT=0
run XK_CAQR_LCHALF=16
run XK_CAQR_LCHALF=4
run XK_CAQR_WT=1
run XK_CAQR_M32=0
run XK_CAQR_OVERLAP=0
