#!/bin/bash
cd $GRAFT_REPO_ROOT
source <(sed -n '/^run()/,/^}/p' tools/gloo_dryrun.sh)
echo "default"; run
echo strong-entities-counts-xavier; run --exchange counts --weights xavier --no-weak
echo weak; run --scaling weak --exchange counts --weights xavier
echo eager-env; KGE_EAGER_COLLECTIVES=1 run --exchange counts --weights xavier --no-weak
echo transh; run --workload transh_fb15k237 --weights xavier --no-weak
