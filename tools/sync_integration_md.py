#!/usr/bin/env python3
"""INTEGRATION.md section 2 quotes integration/dbot/rb_sensor_mi355x.h VERBATIM (from `#pragma once` on): this rewrites
the quoted block from the file, tests/test_cpp_shim.py::test_dbot_binding_compiles asserts the two agree."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "integration", "dbot", "rb_sensor_mi355x.h")).read()
body = src[src.index("#pragma once"):].strip()
path = os.path.join(ROOT, "INTEGRATION.md")
md = open(path).read()
begin, end = "<!-- binding:begin -->", "<!-- binding:end -->"
block = f"{begin}\n```cpp\n// dbot/model/rb_sensor_mi355x.h  (new file in dbot = integration/dbot/rb_sensor_mi355x.h of this repository)\n{body}\n```\n{end}"
assert begin in md and end in md, "INTEGRATION.md lacks the binding markers"
md = re.sub(re.escape(begin) + r".*?" + re.escape(end), lambda m: block, md, flags=re.S)
open(path, "w").write(md)
print("INTEGRATION.md section 2 synchronised:", len(body.splitlines()), "lines")
