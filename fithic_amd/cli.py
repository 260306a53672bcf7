"""`fithic` command line on the MI355X engine: same flags, same banner, same files as the reference's main()
(fithic/fithic.py:43-379).  Quirks kept on purpose: zero means "unset" for -p -b -m -U -L -tL -tU
(fithic.py:194-221, 257-260); an un-gzipped interactions file only prints a message (:155-159); exit status 2
on invalid input (:141-263); -x interOnly never runs extra passes (:349-351)."""
import argparse
import gzip
import os
import sys
import time

from . import fithic as F
from ._version import __version__


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Check the help flag", prog="fithic")
    parser.add_argument("-i", "--interactions", dest="intersfile", required=True,
                        help="REQUIRED: interactions between fragment pairs are read from INTERSFILE")
    parser.add_argument("-f", "--fragments", dest="fragsfile", required=True,
                        help="REQUIRED: midpoints (or start indices) of the fragments are read from FRAGSFILE")
    parser.add_argument("-o", "--outdir", dest="outdir", required=True, help="REQUIRED: where the output files will be written")
    parser.add_argument("-r", "--resolution", dest="resolution", type=int, required=True,
                        help="REQUIRED: resolution of the fixed-size dataset; 0 if the data is not fixed size")
    parser.add_argument("-t", "--biases", dest="biasfile", required=False,
                        help="RECOMMENDED: biases calculated by ICE or KR norm for each locus are read from BIASFILE")
    parser.add_argument("-p", "--passes", dest="noOfPasses", type=int, required=False, help="OPTIONAL: number of spline passes. Default is 1")
    parser.add_argument("-b", "--noOfBins", dest="noOfBins", type=int, required=False,
                        help="OPTIONAL: number of equal-occupancy (count) bins. Default is 100")
    parser.add_argument("-m", "--mappabilityThres", dest="mappabilityThreshold", type=int, required=False,
                        help="OPTIONAL: minimum number of hits per locus that has to exist to call it mappable. DEFAULT is 1.")
    parser.add_argument("-l", "--lib", dest="libname", required=False, help="OPTIONAL: name of the library, used as file prefix. DEFAULT is FitHiC")
    parser.add_argument("-U", "--upperbound", dest="distUpThres", type=int, required=False,
                        help="OPTIONAL: upper bound on the intra-chromosomal distance range (bp). DEFAULT no limit.")
    parser.add_argument("-L", "--lowerbound", dest="distLowThres", type=int, required=False,
                        help="OPTIONAL: lower bound on the intra-chromosomal distance range (bp). DEFAULT no limit.")
    parser.add_argument("-v", "--visual", action="store_true", dest="visual", required=False, help="OPTIONAL: use this flag for generating plots. DEFAULT is False.")
    parser.add_argument("-x", "--contactType", dest="contactType", required=False,
                        help="OPTIONAL: which chromosomal regions to study (intraOnly, interOnly, All). DEFAULT is intraOnly")
    parser.add_argument("-tL", "--biasLowerBound", dest="biasLowerBound", type=float, required=False,
                        help="OPTIONAL: lower bound of bias values to discard. DEFAULT is 0.5")
    parser.add_argument("-tU", "--biasUpperBound", dest="biasUpperBound", type=float, required=False,
                        help="OPTIONAL: upper bound of bias values to discard. DEFAULT is 2")
    parser.add_argument("-V", "--version", action="version", version="Fit-Hi-C {} (fithic-mi355x)".format(__version__),
                        help="Print version and exit")
    parser.add_argument("--device", dest="device", type=int, default=0, help="GPU ordinal (engine option, not in the reference)")
    parser.add_argument("--gpus", dest="gpus", type=int, default=1,
                        help="engine option, not in the reference: shard the contact rows (parts of the file, else by chromosome) over this many GPUs of the "
                             "node (RCCL for the genome-wide steps); the output files are the ones a single GPU writes")
    parser.add_argument("--totals", dest="totals", choices=("reference", "wide"), default="reference",
                        help="engine option, not in the reference: what the binomial is given when the sum of in-range (or inter-chromosomal) "
                             "counts reaches 2^31. 'reference' (default) narrows it to a C int exactly as scipy.special.bdtrc does under "
                             "fithic.py, so the output is the reference's bit for bit - nan p/q for totals in [2^31, 2^32); 'wide' uses the "
                             "true total. A line on stderr says which ran whenever the two differ")
    return parser.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    print("\n")
    print("GIVEN FIT-HI-C ARGUMENTS")
    print("=========================")
    fragsFile = args.fragsfile
    if os.path.exists(fragsFile):
        print("Reading fragments file from: %s" % fragsFile)
    else:
        print("Fragment file not found")
        sys.exit(2)
    try:
        with gzip.open(fragsFile, "r") as f:
            f.readline()
    except Exception:
        print("Fragments file is not gzipped. Exiting now...")
        sys.exit(2)
    contactCountsFile = args.intersfile
    if os.path.isfile(contactCountsFile):
        print("Reading interactions file from: %s" % contactCountsFile)
    else:
        print("Interaction file not found")
        sys.exit(2)
    try:
        with gzip.open(contactCountsFile, "r") as f:
            f.readline()
    except Exception:
        print("Interactions file is not gzipped. Exiting now...")
    outputPath = args.outdir
    if not os.path.isdir(outputPath):
        os.makedirs(outputPath)
        print("Output path created %s" % outputPath)
    else:
        print("Output path being used from %s" % outputPath)
    resolution = args.resolution
    if resolution == 0:
        print("Fixed size data not being used.")
    elif resolution > 0:
        print("Fixed size option detected... Fast version of FitHiC will be used")
        print("Resolution is %s kb" % (resolution / 1000))
    else:
        print("INVALID RESOLUTION ARGUMENT DETECTED")
        print("Please make sure the given resolution is a positive number greater than zero")
        print("User-given resolution: %s" % resolution)
        sys.exit(2)
    if args.biasfile is not None:
        if os.path.isfile(args.biasfile):
            print("Reading bias file from: %s" % args.biasfile)
        else:
            print("Bias file not found")
            sys.exit(2)
    else:
        print("No bias file")
    biasFile = args.biasfile
    noOfPasses = args.noOfPasses if args.noOfPasses else 1
    print("The number of spline passes is %s" % noOfPasses)
    noOfBins = args.noOfBins if args.noOfBins else 100
    print("The number of bins is %s" % noOfBins)
    F.mappThres = args.mappabilityThreshold if args.mappabilityThreshold else 1
    print("The number of reads required to consider an interaction is %s" % F.mappThres)
    libName = args.libname if args.libname else "FitHiC"
    print("The name of the library for outputted files will be %s" % libName)
    F.distUpThres = args.distUpThres if args.distUpThres else float("inf")
    F.distLowThres = args.distLowThres if args.distLowThres else 0
    print("Upper Distance threshold is %s" % F.distUpThres)
    print("Lower Distance threshold is %s" % F.distLowThres)
    F.visual = False
    if args.visual:
        print("Graphs will be outputted")
        F.visual = True
    region = args.contactType if args.contactType is not None else "intraOnly"
    F.interOnly = F.allReg = False
    if region == "All":
        print("All genomic regions will be analyzed")
        F.allReg = True
    elif region == "interOnly":
        print("Only inter-chromosomal regions will be analyzed")
        F.interOnly = True
    elif region == "intraOnly":
        print("Only intra-chromosomal regions will be analyzed")
    else:
        print("Invalid Option. Only options are 'All', 'interOnly', or 'intraOnly'")
        sys.exit(2)
    F.biasLowerBound = args.biasLowerBound if args.biasLowerBound else 0.5
    F.biasUpperBound = args.biasUpperBound if args.biasUpperBound else 2
    if F.biasLowerBound > F.biasUpperBound:
        print("Invalid Option. Bias lower bound is greater than bias upper bound. Please fix.")
        sys.exit(2)
    print("Lower bound of bias values is %s" % F.biasLowerBound)
    print("Upper bound of bias values is %s" % F.biasUpperBound)
    print("All arguments processed. Running FitHiC now...")
    print("=========================")
    print("\n")

    F.reset_session()
    try:                                       # the engine (and, with --gpus N, its worker processes) is released on every path
        F.resolution = resolution
        F.device = args.device
        if args.gpus < 1:
            print("Invalid Option. --gpus must be at least 1")
            sys.exit(2)
        F.gpus = args.gpus
        F.totals = args.totals
        F.logfile = os.path.join(outputPath, libName + ".fithic.log")

        (mainDic, observedInterAllCount, observedInterAllSum, observedIntraAllSum, observedIntraInRangeSum) = \
            F.read_Interactions(contactCountsFile, biasFile)
        binStats = F.makeBinsFromInteractions(mainDic, noOfBins, observedIntraInRangeSum)
        (binStats, noOfFrags, maxPossibleGenomicDist, possibleIntraInRangeCount, possibleInterAllCount, interChrProb,
         baselineIntraChrProb) = F.generate_FragPairs(observedInterAllCount, observedInterAllSum, binStats, fragsFile, resolution)
        biasDic = F.read_biases(biasFile) if biasFile else 0
        (x, y, yerr) = F.calculateProbabilities(mainDic, binStats, resolution, os.path.join(outputPath, libName + ".fithic_pass1"),
                                                observedIntraInRangeSum)
        t_first = time.time()
        print("Spline fit Pass 1 starting...")
        outliersline, outliersdist = [], []
        (splineXinit, splineYinit, residual, outliersline, outliersdist, FDRXinit, FDRYinit) = F.fit_Spline(
            mainDic, x, y, yerr, contactCountsFile, os.path.join(outputPath, libName + ".spline_pass1"), biasDic, outliersline,
            outliersdist, observedIntraInRangeSum, possibleIntraInRangeCount, possibleInterAllCount, observedInterAllCount,
            observedIntraAllSum, observedInterAllSum, F.biasLowerBound, F.biasUpperBound, resolution, 1)
        print("Number of outliers is... %s" % len(outliersline))
        t_first_end = time.time()
        print("Spline fit Pass 1 completed. Time took %s" % (t_first_end - t_first))

        for i in range(2, 1 + noOfPasses):
            if F.interOnly:
                print("Extra spline fits will not help with interOnly spline fit... Bypassing option")
                break
            print("\n")
            print("\n")
            (mainDic, observedInterAllCount, observedInterAllSum, observedIntraAllSum, observedIntraInRangeSum) = \
                F.read_Interactions(contactCountsFile, biasFile, outliersline)
            binStats = F.makeBinsFromInteractions(mainDic, noOfBins, observedIntraInRangeSum, outliersdist)
            (binStats, noOfFrags, maxPossibleGenomicDist, possibleIntraInRangeCount, possibleInterAllCount, interChrProb,
             baselineIntraChrProb) = F.generate_FragPairs(observedInterAllCount, observedInterAllSum, binStats, fragsFile, resolution)
            (x, y, yerr) = F.calculateProbabilities(mainDic, binStats, resolution, os.path.join(outputPath, libName + ".fithic_pass" + str(i)),
                                                    observedIntraInRangeSum)
            print("Spline fit Pass %s starting..." % i)
            (splineX, splineY, residual, outliersline, outliersdist, FDRX, FDRY) = F.fit_Spline(
                mainDic, x, y, yerr, contactCountsFile, os.path.join(outputPath, libName + ".spline_pass" + str(i)), biasDic, outliersline,
                outliersdist, observedIntraInRangeSum, possibleIntraInRangeCount, possibleInterAllCount, observedInterAllCount,
                observedIntraAllSum, observedInterAllSum, F.biasLowerBound, F.biasUpperBound, resolution, i)
            print("Spline fit Pass %s completed. Time took %s" % (i, (t_first_end - t_first)))   # the reference prints pass 1's time (:372)
            if F.visual:
                from . import plots
                plots.compare_Spline_FDR(FDRXinit, FDRYinit, FDRX, FDRY, os.path.join(outputPath, libName + ".spline_FDR_comparison"), str(i))
                plots.compareFits_Spline(splineXinit, splineYinit, splineX, splineY, os.path.join(outputPath, libName + ".spline_comparison"), str(i))
        print("=========================")
        print("Fit-Hi-C completed successfully")
        print("\n")
    finally:
        stuck = F.session_stuck()
        F.reset_session()
        F.gpus = 1                             # a --gpus N of this call does not outlive it (callers of the stage functions set their own)
        if stuck:
            # --gpus N lost a rank while this process's own rank sat in a collective: that thread can never return, and a normal
            # interpreter exit (HIP / RCCL teardown) would wait for it.  Say what happened and leave.
            import traceback
            traceback.print_exc()
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(1)


if __name__ == "__main__":
    main()
