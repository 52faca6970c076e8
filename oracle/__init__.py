"""CPU oracle for the U-Net / PHiSeg / Probabilistic-U-Net hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and there only as the *checker*.  The
product path (``unet-zoo_amd``) never imports this package and fails loudly
when the HIP library is missing.

What it is: a plain fp32 PyTorch (CPU, ``torch.nn.functional``) restatement of
the reference's op graph, written functionally over a ``state_dict``-keyed
parameter dictionary, with the latent noise eps as an explicit input.  Each
function cites the reference file:line it restates.

Second part, plain C (``oracle/uz_cpu.c`` -> ``oracle/_build/libuz_cpu.so``, built
by ``make -C oracle`` / ``__graft_entry__.build()``): scalar ``uz_cpu_*`` twins of the
C-ABI entry points with the SAME argument lists minus the stream,
checked against ``torch.nn.functional`` on the CPU and used as the checker of the HIP
kernels through identical calls (``tests/test_cpu_twins.py``).

Pinning: the reference has no tests or golden vectors for this path
(SURVEY.md section 4 / 8c), so the oracle is pinned against outputs of the
reference itself, generated in the build container by
``tools/gen_golden.py`` (imports ``/root/reference`` read-only) and committed
as data under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks the
oracle against every one of them.

One exception, PARITY UNPINNED: the reversible blocks (``rev_sequence``) wrap
``revtorch==0.2.0`` (requirements.txt:38), which is neither vendored under
/root/reference nor installable here; its published additive-coupling algorithm
is restated and no reference output pins it.
"""
from .refgraph import (  # noqa: F401
    phiseg_forward, phiseg_loss, phiseg_accumulate_output, phiseg_eps_shapes,
    unet_forward, unet_loss,
    probunet_forward, probunet_loss, probunet_fcomb,
    kl_two_gauss_with_diag_cov, batch_to_onehot,
    adam_reference_step, synthetic_batch, deterministic_state_dict,
    rev_sequence, revtorch_second_bn_update,
)
