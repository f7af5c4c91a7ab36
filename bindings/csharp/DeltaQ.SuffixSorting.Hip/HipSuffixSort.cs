// HipSuffixSort.cs -- C# host shim: an ISuffixSort provider whose body is a P/Invoke into
// libdq_sufsort_hip.so (C ABI: include/dq_sufsort.h).  Drop-in for
//   new LibDivSufSort()   (src/DeltaQ.SuffixSorting.LibDivSufSort/LibDivSufSort.cs:10-32)
// wherever an ISuffixSort is accepted: Diff.Create (src/DeltaQ.BsDiff/Diff.cs:27,89-90), the
// dq CLI (src/DeltaQ.CommandLine/Commands.BsDiff.cs:30-35), tests and benchmarks.
//
// This file ships as SOURCE: the build image has no dotnet SDK, so it has not been compiled
// here.  The tested surface is the C ABI it binds (tests/test_gpu_parity.py goes through the
// same entry points via ctypes).
using CommunityToolkit.HighPerformance.Buffers;
using System;
using System.Buffers;
using System.Runtime.InteropServices;

namespace DeltaQ.SuffixSorting.Hip;

/// <summary>
/// Suffix sorting on an AMD Instinct MI355X through libdq_sufsort_hip.
/// The returned suffix array is bit-identical to <c>LibDivSufSort.Sort</c>.
/// </summary>
public sealed class HipSuffixSort : ISuffixSort
{
    private const string Lib = "dq_sufsort_hip";          // libdq_sufsort_hip.so on the probing path

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
    private static extern unsafe int dq_sufsort_hip_i32(byte* text, long n, int* sa, int device);

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
    private static extern int dq_abi_version();

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
    private static extern int dq_device_count();

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
    private static extern IntPtr dq_last_error();

    private readonly int _device;
    private readonly ISuffixSort? _fallback;

    /// <param name="device">HIP device ordinal; -1 = DQ_HIP_DEVICE or device 0.</param>
    /// <param name="fallback">
    /// Provider that takes over when the native call fails (no device, out of device memory, text beyond the
    /// native limits).  Default: none -- a failure throws <see cref="InvalidOperationException"/> -- unless the
    /// environment variable DQ_HIP_FALLBACK is set to "divsufsort", which installs the managed
    /// <c>LibDivSufSort</c> (SURVEY section 8(b), "Preconditions").  The native library itself never falls back.
    /// </param>
    public HipSuffixSort(int device = -1, ISuffixSort? fallback = null)
    {
        _device = device;
        _fallback = fallback;
        if (_fallback is null
            && string.Equals(Environment.GetEnvironmentVariable("DQ_HIP_FALLBACK"), "divsufsort", StringComparison.OrdinalIgnoreCase))
        {
            _fallback = new DeltaQ.SuffixSorting.LibDivSufSort.LibDivSufSort();
        }

        try
        {
            if (dq_abi_version() != 1)
            {
                throw new InvalidOperationException("libdq_sufsort_hip ABI version mismatch");
            }
        }
        catch (DllNotFoundException) when (_fallback is not null)
        {
            _nativeMissing = true;            // every call goes to the fallback
        }
    }

    private readonly bool _nativeMissing;

    public static int DeviceCount => dq_device_count();

    // ISuffixSort.cs:18 -- same allocation behaviour as LibDivSufSort.cs:14 (pooled, uncleared)
    public IMemoryOwner<int> Sort(ReadOnlySpan<byte> textBuffer)
    {
        var owner = MemoryOwner<int>.Allocate(textBuffer.Length);
        try
        {
            SortCore(textBuffer, owner.Span);
            return owner;
        }
        catch
        {
            owner.Dispose();
            throw;
        }
    }

    // ISuffixSort.cs:27 -- same precondition and message as LibDivSufSort.cs:23-31
    public void Sort(ReadOnlySpan<byte> textBuffer, Span<int> suffixBuffer)
    {
        if (textBuffer.Length != suffixBuffer.Length)
        {
            ThrowHelper();
        }

        SortCore(textBuffer, suffixBuffer);
    }

    private unsafe void SortCore(ReadOnlySpan<byte> text, Span<int> sa)
    {
        // n = 0 hands the native side null pointers, which it accepts (no-op), like
        // DivSufSort.cs:24.  Exactly text.Length ints are written: Diff.Create's I[n]
        // (Diff.cs:78,89-90) is never touched.
        if (_nativeMissing)
        {
            _fallback!.Sort(text, sa);
            return;
        }

        int rc;
        fixed (byte* pText = text)
        fixed (int* pSa = sa)
        {
            rc = dq_sufsort_hip_i32(pText, text.Length, pSa, _device);
        }

        if (rc == 0)
        {
            return;
        }

        // -1 bad arguments is a caller bug and never retried; -2 OOM, -3 HIP error, -4 too large, -5 no device are
        // what a fallback is for (include/dq_sufsort.h).  The native side has written nothing it did not finish.
        if (_fallback is not null && rc != -1)
        {
            _fallback.Sort(text, sa);
            return;
        }

        string msg = Marshal.PtrToStringAnsi(dq_last_error()) ?? string.Empty;
        throw new InvalidOperationException($"dq_sufsort_hip_i32 failed ({rc}): {msg}");
    }

    private static void ThrowHelper() => throw new ArgumentException("Text and suffix buffers should have the same length");
}
