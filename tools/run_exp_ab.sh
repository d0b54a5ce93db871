#!/bin/bash
# usage: tools/run_exp_ab.sh <library>: kernel traces of the bench step with the in-tree library and with <library>
cd "$(dirname "$0")/.." || exit 1
tools/prof_step.sh base || exit 1
VITK_LIB=$PWD/$1 tools/prof_step.sh exp || exit 1
