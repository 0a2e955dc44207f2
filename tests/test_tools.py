"""CPU: the benchmark-log scraper (tools/scrape_bench_log.py) on a log in the reference's shape."""
import importlib.util
import os

from conftest import ROOT

LOG = """+ ./ci/run something
name, driver_version
AMD Instinct MI355X (gfx950), 6.12.12
Model name:                         AMD EPYC 9575F 64-Core Processor
Core(s) per socket:                 64
hostname:box17
compiler:hipcc-7.2-gfx950
algorithm,dim,precision,nsteps,nbodies,total [s]
all-pairs,3,64,190,100000,1.19
compiler:hipcc-7.2-gfx950
algorithm,dim,precision,nsteps,nbodies,total [s],force [s],accel [s],bbox [s],sort [s],multipoles [s],force approx [s]
bvh,3,64,1000,100000,1.32,1.30,0.02,0.05,0.31,0.12,0.80
sequential
compiler:gcc
octree,3,64,190,10000,9.50
"""


def _mod():
    spec = importlib.util.spec_from_file_location("scrape_bench_log", os.path.join(ROOT, "tools", "scrape_bench_log.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_scraper_columns_and_rows():
    m = _mod()
    header, rows = m.scrape(LOG.splitlines(True))
    text = m.to_csv(header, rows).splitlines()
    assert text[0].startswith("gpu,driver,cpu,#cores,seq,compiler,hostname,algorithm,dim,precision,nsteps,nbodies,total [s],force [s]")
    assert text[1].startswith("AMD Instinct MI355X (gfx950),6.12.12,AMD EPYC 9575F 64-Core Processor,64,False,hipcc-7.2-gfx950,box17,all-pairs,3,64,190,100000,1.19")
    assert text[2].split(",")[7:] == "bvh,3,64,1000,100000,1.32,1.30,0.02,0.05,0.31,0.12,0.80".split(",")
    assert text[3].split(",")[4:8] == ["True", "gcc", "box17", "octree"]
    assert all(len(t.split(",")) == len(text[0].split(",")) for t in text)
