"""Shared pieces of the data plugins (host side)."""
import argparse


def default_rawboost_args():
    """The RawBoost hyper-parameter defaults of the reference CLI (main.py:258-298)."""
    return argparse.Namespace(algo=5, nBands=5, minF=20, maxF=8000, minBW=100, maxBW=1000, minCoeff=10, maxCoeff=100, minG=0,
                              maxG=0, minBiasLinNonLin=5, maxBiasLinNonLin=20, N_f=5, P=10, g_sd=2, SNRmin=10, SNRmax=40)
