#!/bin/bash
# old vs new library over batch sizes (automatic split scheme): python tools/scheme_table.py prints both schemes
for v in old new; do echo "== $v"; PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/scheme_table.py 2>/dev/null; done
