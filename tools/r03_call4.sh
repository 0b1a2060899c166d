#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
LAB_TAG=_d LAB_ARGS="--only shipped,256x256,pln_pln_128x128_bk16_nb3_p,f32k_pln_128x128_bk32_nb2_p,f32k_f32k_256x128,f32k_f32k_128x128_bk32_nb2_p" bash tools/lab/gemm3_round.sh
