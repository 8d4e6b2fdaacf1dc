cd $GRAFT_REPO_ROOT
python tools/brick_cfg_ab.py LEGO - bwd_brick=1 bwd_brick=1,bwd_item12=0 bwd_brick=1,bwd_brick_hi=8 bwd_brick=1,bwd_brick_hi=6 2>/dev/null
R3_DTYPE=f16 python tools/brick_cfg_ab.py LEGO - bwd_brick=1 bwd_brick=1,bwd_item12=0 bwd_brick=1,bwd_brick_hi=8 2>/dev/null
R3_DTYPE=f16 python tools/brick_cfg_ab.py S1 - bwd_brick=1 bwd_brick=1,bwd_item12=0 2>/dev/null
