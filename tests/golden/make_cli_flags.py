#!/usr/bin/env python3
"""Generate tests/golden/cli_flags.json: the flags of the reference's `assemble`, `call` and `call-exact` programs -- name,
arity, type and default of every entry of ASSEMBLE_MCMC_PARSER_ARGUMENTS / CALL_MCMC_PARSER_ARGUMENTS /
CALL_EXACT_PARSER_ARGUMENTS (application/arguments.py:742-838) -- by importing the reference under the identity shims of
tests/golden/_shim.  Build container only (needs /root/reference); the output is data (a table of flag names and defaults),
not source.  Usage: python tests/golden/make_cli_flags.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "_shim"))
sys.path.insert(0, "/root/reference")

from mchap.application import arguments as A  # noqa: E402

tables = {"assemble": A.ASSEMBLE_MCMC_PARSER_ARGUMENTS, "call": A.CALL_MCMC_PARSER_ARGUMENTS, "call-exact": A.CALL_EXACT_PARSER_ARGUMENTS}
out = {}
for prog, table in tables.items():
    rows = []
    for arg in table:
        kw = arg.kwargs
        if isinstance(arg, A.BooleanFlag):
            rows.append(dict(flag=arg.cli, kind="flag", dest=kw["dest"], action=kw["action"]))
        else:
            t = kw.get("type")
            rows.append(dict(flag=arg.cli, kind="parameter", nargs=kw.get("nargs"), type=getattr(t, "__name__", None), default=kw.get("default")))
    out[prog] = rows
with open(os.path.join(HERE, "cli_flags.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print({k: len(v) for k, v in out.items()})
