"""The reference's `-v` figures (fithic/fithic.py:970-999, 1256-1263, 1267-1321), drawn on the host with matplotlib from
the arrays the engine already produced: bin means and the spline table (host fit), the 51 FDR-threshold counts (device
histogram of q, `k_fdr_hist`).  Nothing here touches the GPU; the figures carry the same series, labels, limits and
file names as the reference's.  matplotlib is imported lazily so that a run without -v never needs it.
"""
import numpy as np

toKb = 10 ** -3        # fithic/fithic.py:38-41
toProb = 10 ** 5


def _plt():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from matplotlib.ticker import MaxNLocator
    return plt, MaxNLocator


def _scaled(values, k):
    return [v * k for v in values]


def plot_spline_fit(outfilename, passNo, x, y, yerr, splineX, newSplineY, distLowThres, distUpThres):
    """fithic/fithic.py:970-999: bin means with the fitted (monotone) spline, linear and log-log panels."""
    plt, MaxNLocator = _plt()
    print("Plotting %s" % (outfilename + ".png"))
    plt.clf()
    fig = plt.figure()
    ax = fig.add_subplot(2, 1, 1)
    plt.plot(_scaled(splineX, toKb), _scaled(newSplineY, toProb), "g-", label="spline-" + str(passNo), linewidth=2)
    plt.errorbar(_scaled(x, toKb), _scaled(y, toProb), _scaled(yerr, toProb), fmt="r.", label="Mean with std. error", linewidth=2)
    plt.ylabel("Contact probability (x10$^{-5}$)")
    plt.xlabel("Genomic distance (kb)")
    bounded = distLowThres > 0 and distUpThres < float("inf")
    if bounded:
        plt.xlim(_scaled([distLowThres, distUpThres], toKb))
    plt.gca().yaxis.set_major_locator(MaxNLocator(nbins=3, prune=None))
    ax.legend(loc="upper right")
    fig.add_subplot(2, 1, 2)
    plt.loglog(splineX, newSplineY, "g-")
    plt.errorbar(x, y, yerr=yerr, fmt="r.")
    if bounded:
        plt.xlim([distLowThres, distUpThres])
    plt.ylabel("Contact probability (log-scale)")
    plt.xlabel("Genomic distance (log-scale)")
    plt.savefig(outfilename + ".png")
    plt.close(fig)


def plot_qvalues(qvalTicks, significantTicks, outfilename):
    """fithic/fithic.py:1256-1263: significant contacts per FDR threshold (the counts come from the device)."""
    plt, _ = _plt()
    plt.clf()
    fig = plt.figure()
    fig.add_subplot(1, 1, 1)
    plt.plot(qvalTicks, significantTicks, "b*-")
    plt.xlabel("FDR threshold")
    plt.ylabel("Number of significant contacts")
    plt.savefig(outfilename + ".png")
    plt.close(fig)


def compare_Spline_FDR(splineFDRxinit, splineFDRyinit, splineFDRx, splineFDRy, figname, i):
    """fithic/fithic.py:1267-1280."""
    plt, MaxNLocator = _plt()
    plt.clf()
    fig = plt.figure()
    ax = fig.add_subplot(1, 1, 1)
    plt.plot(splineFDRx[1:], _scaled(splineFDRy[1:], toKb), "r+-", label="spline-" + str(i))
    plt.plot(splineFDRxinit[1:], _scaled(splineFDRyinit[1:], toKb), "g.-", label="spline-1")
    plt.xlabel("FDR threshold")
    plt.ylabel("Significant contacts (x10$^{3}$)")
    plt.gca().yaxis.set_major_locator(MaxNLocator(prune="lower"))
    lg = ax.legend(loc="lower right")
    lg.draw_frame(False)
    plt.savefig(figname + ".png")
    plt.close(fig)


def compareFits_Spline(splineXinit, splineYinit, splineX, splineY, figname, X):
    """fithic/fithic.py:1282-1321: the first and the current spline, each down-sampled to <= 5000 random table points."""
    plt, MaxNLocator = _plt()
    downsample = min(5000, len(splineXinit))
    plt.clf()
    fig = plt.figure()
    ax = fig.add_subplot(1, 1, 1)

    def sample(xs, ys):
        idx = sorted(np.random.choice(len(xs), downsample))
        return _scaled([xs[j] for j in idx], toKb), _scaled([ys[j] for j in idx], toProb)

    x, y = sample(splineXinit, splineYinit)
    plt.plot(x, y, "g.-", label="spline-1")
    if figname[-1] != "1":
        x, y = sample(splineX, splineY)
        plt.plot(x, y, "r.-", label="spline-" + X)
    elif max(x) > 1000:
        plt.xlim([500, 1000])
        plt.ylim([0, 1.0])
    else:
        plt.xlim([50, 100])
        plt.ylim([0, 0.5])
    ax.legend(loc="upper right")
    plt.xlabel("Genomic distance (kb)")
    plt.ylabel("Contact probability (x10$^{-5}$)")
    plt.gca().yaxis.set_major_locator(MaxNLocator(prune="lower"))
    plt.savefig(figname + ".png")
    plt.close(fig)
