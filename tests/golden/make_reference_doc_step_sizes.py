"""Extracts the reference-held sampler outputs of the reference's FROZEN documentation (``/root/reference/docs/_freeze/*/
execute-results/html.json``: the executed cells of docs/index.qmd, stan-usage.qmd, pymc-usage.qmd) into a small fixture.

The docs run ``nutpie.sample(compiled)`` with default settings (6 chains, tune 400, draws 1000) and the rendered progress table
shows, per chain, the FINAL step size and the gradient evaluations of the last draw.  The models are tiny and analytic, so the
numbers can be compared with ensembles of this repository's sampler: they are the only outputs of nuts-rs itself in the
reference's tree besides the three HalfNormal files under tests/reference/.

Run in the build container (reads /root/reference; the GPU box has no such directory):
    python tests/golden/make_reference_doc_step_sizes.py
"""
import json
import os
import re

DOCS = "/root/reference/docs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_doc_step_sizes.json")


def tables(doc):
    md = json.load(open(f"{DOCS}/_freeze/{doc}/execute-results/html.json"))["result"]["markdown"]
    out = []
    for body in re.findall(r'<tbody id="chain-details">(.*?)</tbody>', md, re.S):
        rows = re.findall(r"<td>(\d+)</td>\s*<td>(\d+)</td>\s*<td>([\d.]+)</td>\s*<td>(\d+)</td>", body)
        out.append([{"draws": int(a), "divergences": int(b), "step_size": float(c), "gradients_last_draw": int(d)} for a, b, c, d in rows])
    return out


idx, stan, pymc, stats, nf = tables("index"), tables("stan-usage"), tables("pymc-usage"), tables("sample-stats"), tables("nf-adapt")


def nf_adapt_totals():
    """docs/nf-adapt.qmd:107-122: the printed totals of the run WITHOUT normalizing flows (the second `Number of gradient evaluations` /
    `Minimum effective sample size` pair of the page; the first is the flow run)"""
    md = json.load(open(f"{DOCS}/_freeze/nf-adapt/execute-results/html.json"))["result"]["markdown"]
    pairs = re.findall(r"Number of gradient evaluations: (\d+)\s+Minimum effective sample size: ([\d.]+)", md)
    assert len(pairs) == 2, pairs
    return {"with_flow": {"gradient_evaluations": int(pairs[0][0]), "min_ess": float(pairs[0][1])},
            "diag": {"gradient_evaluations": int(pairs[1][0]), "min_ess": float(pairs[1][1])}}


def ess_values():
    """`az.ess(trace)` of the first regression run (docs/pymc-usage.qmd:105-108): the bulk effective sample sizes of 6 x 1000 draws"""
    import html

    md = json.load(open(f"{DOCS}/_freeze/pymc-usage/execute-results/html.json"))["result"]["markdown"]
    seg = html.unescape(re.sub(r"<[^>]+>", " ", md[md.find("az.ess"):][:40000]))
    return {name: float(re.search(name + r"\s*\(\)\s*float64\s*[\d.e+]+\s*array\(([\d.]+)\)", seg).group(1)) for name in ("intercept", "slope")}


fixture = {
    "_source": "docs/_freeze/{index,stan-usage,pymc-usage,sample-stats,nf-adapt}/execute-results/html.json of the reference (progress tables of the executed cells)",
    "_settings": "nutpie.sample(compiled): 6 chains, tune 400, draws 1000, every other setting default",
    # mu ~ N(0, 1); obs ~ N(mu, 1), observed [1, 2, 3]   (docs/index.qmd:39-46 through PyMC, :66-91 and docs/stan-usage.qmd:57-84 through Stan)
    "normal_1d": {"posterior": "N(1.5, 1/4)", "runs": [idx[0], idx[1], stan[0]], "cites": ["docs/index.qmd:39-46", "docs/index.qmd:66-91", "docs/stan-usage.qmd:57-84"]},
    # intercept, slope ~ N(0, 1); y ~ N(intercept + slope * x, 0.1), x = [1, 2, 3], observed [1, 2, 3]   (docs/pymc-usage.qmd:53-79 and :173-187)
    "regression_x123": {"posterior": "precision [[301, 600], [600, 1401]], X'y / sigma^2 = [600, 1400]", "runs": [pymc[0], pymc[1]], "bulk_ess_of_the_first_run": ess_values(),
                        "cites": ["docs/pymc-usage.qmd:53-79", "docs/pymc-usage.qmd:173-187"]},
    # the same model after with_data(x=[4, 5, 6])   (docs/pymc-usage.qmd:191-194)
    "regression_x456": {"posterior": "precision [[301, 1500], [1500, 7701]], X'y / sigma^2 = [600, 3200]", "runs": [pymc[2]], "cites": ["docs/pymc-usage.qmd:191-194"]},
    # Neal's funnel: log_sigma ~ N(0, 1); x[5] ~ N(0, exp(log_sigma)); nutpie.sample(compiled, tune=1000, seed=42, ...)   (docs/sample-stats.qmd:18-35)
    "funnel_diag": {"settings": "tune 1000, draws 1000, adaptation diag (default)", "runs": [stats[0]], "cites": ["docs/sample-stats.qmd:18-35"]},
    # x ~ N(0, 1); y ~ N(x, 0.01); z[100] ~ N(y, 1)  — 102 dimensions, one very stiff direction; nutpie.sample(compiled, tune=1000, seed=42, ...) with
    # the default ("diag") adaptation   (docs/sample-stats.qmd:141-157; the frozen output is the SECOND progress table of that page: the page's
    # low_rank cell, :256-268, is not in the frozen results)
    "correlated_102d": {"settings": "tune 1000, draws 1000, adaptation diag (default)", "runs": [stats[1]], "cites": ["docs/sample-stats.qmd:141-157"]},
    # log_sigma ~ N(0, 1); x[100] ~ N(0, exp(log_sigma / 2)) — 101 dimensions; nutpie.sample(compiled, seed=1): defaults (6 chains, tune 400, draws 1000,
    # "diag").  The page prints the TOTAL gradient evaluations of the run incl. warm-up and the minimum bulk ESS: the only reference-held number that
    # integrates the whole warm-up (docs/nf-adapt.qmd:60-78, 115-122; the first progress table of the page is this run, the second the flow run)
    "funnel_101d": {"settings": "6 chains, tune 400, draws 1000, adaptation diag (default), seed 1", "runs": [nf[0]], "totals": nf_adapt_totals()["diag"],
                    "cites": ["docs/nf-adapt.qmd:60-78", "docs/nf-adapt.qmd:115-122"]},
}
json.dump(fixture, open(OUT, "w"), indent=1)
print(OUT, {k: [len(r) for r in v["runs"]] for k, v in fixture.items() if not k.startswith("_")})
