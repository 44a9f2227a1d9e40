"""ECM iterations on the bench workload (for rocprofv3 passes); SHARD=8:6 restricts it to one rank's chromosomes, ITERS sets the count."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths
lengths = hg38_chain_lengths(200)
if os.environ.get("SHARD"):
    from consenrich_amd.sharding import lpt_assign
    w, r = map(int, os.environ["SHARD"].split(":")); lengths = [lengths[i] for i in lpt_assign(lengths, w)[r]]
b = DeviceBatch(0); b.configure(ModelParams(state_dim=2), 32, lengths); b.synthesize(1234); b.stats()
b.ecm(max_iters=int(os.environ.get("ITERS", "2")), inner_iters=5, rtol=0.0, use_lambda=False, use_kappa=True)
b.synchronize()
