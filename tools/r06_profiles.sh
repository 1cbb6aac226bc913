#!/bin/bash
# Runs on the GPU box (via gpurun): the round's profile set -- kernel trace + FETCH_SIZE pass (lec_ kernels only) for the five bench
# configurations; tools/summarize_prof.py r06 <tag> <key> turns gpurun_out/prof_<tag>/ into profiles/r06_<tag>_*.
R=${GRAFT_REPO_ROOT:-$(pwd)}
tools/prof.sh all "FETCH_SIZE GRBM_GUI_ACTIVE" --steps 10 --warmup 2 &&
tools/prof.sh noq "FETCH_SIZE GRBM_GUI_ACTIVE" --no-q --steps 10 --warmup 2 &&
tools/prof.sh f32 "FETCH_SIZE GRBM_GUI_ACTIVE" --storage f32 --steps 10 --warmup 2 &&
tools/prof.sh moving "FETCH_SIZE GRBM_GUI_ACTIVE" --moving --timesteps 512 --steps 10 --warmup 2 &&
tools/prof.sh moving2048 "FETCH_SIZE GRBM_GUI_ACTIVE" --moving --timesteps 2048 --steps 10 --warmup 2 &&
tools/prof.sh moving_cube "FETCH_SIZE GRBM_GUI_ACTIVE" --moving --timesteps 512 --moving-layout cube --steps 10 --warmup 2
