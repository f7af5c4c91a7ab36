// HipDiff.cs -- C# host shim for the rows next to the suffix sort (SURVEY.md section 8(f)): the match search on the
// device-resident suffix array, Diff.Create / Patch.Apply natively, and "one old file, many new files".
// P/Invoke into libdq_sufsort_hip.so; every entry point is declared in include/dq_sufsort.h, which cites the
// reference line each one replaces:
//   HipDiff.Create        Diff.Create(oldData, newData, output, suffixSort)   src/DeltaQ.BsDiff/Diff.cs:27-253
//   HipDiff.Apply         Patch.Apply(input, diff, output)                    src/DeltaQ.BsDiff/Patch.cs:34-43,52-168
//   HipMatchSearch.Search Search(I, oldData, newData[scan..], 0, n, out pos)  src/DeltaQ.BsDiff/Diff.cs:267-298
//   HipDiffIndex          Diff.cs:89-90 paid once per old file
//
// Ships as SOURCE (no dotnet SDK in the build image); the tested surface is the C ABI (tests/test_gpu_bsdiff.py,
// tests/test_gpu_match_search.py bind the same exports through ctypes).
using System;
using System.IO;
using System.Runtime.InteropServices;

namespace DeltaQ.SuffixSorting.Hip;

internal static unsafe class Native
{
    internal const string Lib = "dq_sufsort_hip";

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
    internal static extern IntPtr dq_last_error();

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
    internal static extern int dq_bsdiff_search_i32(byte* oldData, long n, int* sa, byte* newData, long m, long* scans,
                                                    long scan0, long count, long cap, int* pos, int* len, int device);

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
    internal static extern int dq_bsdiff_create(byte* oldData, long n, byte* newData, long m, byte* patch, long cap,
                                                long* patchLen, int device);

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
    internal static extern long dq_bsdiff_patch_bound(long n, long m);

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
    internal static extern int dq_bspatch_apply(byte* oldData, long n, byte* patch, long patchLen, byte* output, long cap,
                                                long* outLen);

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
    internal static extern int dq_bsdiff_index_create(byte* oldData, long n, void* dOld, void* dSa, int device, IntPtr* index);

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
    internal static extern int dq_bsdiff_index_diff(IntPtr index, byte* newData, long m, byte* patch, long cap, long* patchLen);

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
    internal static extern int dq_bsdiff_index_clone(IntPtr index, int device, IntPtr* indexOut);

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
    internal static extern void dq_bsdiff_index_free(IntPtr index);

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
    internal static extern int dq_last_diff_info(long* info, int count);

    internal static string LastError() => Marshal.PtrToStringAnsi(dq_last_error()) ?? string.Empty;

    internal static void Check(int rc, string what)
    {
        if (rc != 0)
        {
            throw new InvalidOperationException($"{what} failed ({rc}): {LastError()}");
        }
    }
}

/// <summary>Diff.Create / Patch.Apply with the suffix sort, the match search and the bzip2 block sorts on the MI355X.</summary>
public static class HipDiff
{
    /// <summary>
    /// Same contract as <c>Diff.Create(oldData, newData, output, suffixSort)</c> (Diff.cs:27-52): the stream must be
    /// writable and seekable; the patch is BSDIFF40 and is read by <c>Patch.Apply</c> of either implementation.
    /// </summary>
    public static unsafe void Create(ReadOnlySpan<byte> oldData, ReadOnlySpan<byte> newData, Stream output, int device = -1)
    {
        if (output is null)
        {
            throw new ArgumentNullException(nameof(output));
        }

        if (!output.CanSeek)
        {
            throw new ArgumentException("Output stream must be seekable.", nameof(output));
        }

        if (!output.CanWrite)
        {
            throw new ArgumentException("Output stream must be writable.", nameof(output));
        }

        byte[] patch = CreateBytes(oldData, newData, device, out long length);
        output.Write(patch, 0, checked((int)length));
    }

    /// <summary>
    /// Shape of the last Create / HipDiffIndex.Create on this thread (dq_last_diff_info): Search calls of the loop
    /// (Diff.cs:106), windows, positions asked again exactly, device scans given back to the host loop, workgroups.
    /// A non-zero fourth entry means the call was correct but slow: the device was kept full by other work.
    /// Entries 5..8 (several grids on one new file): grids launched, grids the followed one was joined to, grids dropped
    /// unjoined, control triples taken over from the grids' own emitter threads.
    /// </summary>
    public static unsafe long[] LastDiffInfo()
    {
        var info = new long[9];
        fixed (long* p = info)
        {
            Native.Check(Native.dq_last_diff_info(p, info.Length), nameof(Native.dq_last_diff_info));
        }

        return info;
    }

    public static unsafe byte[] CreateBytes(ReadOnlySpan<byte> oldData, ReadOnlySpan<byte> newData, int device, out long length)
    {
        long cap = Native.dq_bsdiff_patch_bound(oldData.Length, newData.Length);
        var patch = new byte[cap];
        long len = 0;
        fixed (byte* pOld = oldData)
        fixed (byte* pNew = newData)
        fixed (byte* pPatch = patch)
        {
            Native.Check(Native.dq_bsdiff_create(pOld, oldData.Length, pNew, newData.Length, pPatch, cap, &len, device),
                         nameof(Native.dq_bsdiff_create));
        }

        length = len;
        return patch;
    }

    /// <summary>
    /// <c>Patch.Apply(input, diff, output)</c> (Patch.cs:34-43).  A patch the reference rejects raises the same
    /// <see cref="InvalidOperationException"/>("Corrupt patch").  Host code only: no device is needed.
    /// </summary>
    public static unsafe void Apply(ReadOnlySpan<byte> input, ReadOnlySpan<byte> diff, Stream output)
    {
        if (output is null)
        {
            throw new ArgumentNullException(nameof(output));
        }

        long newSize = 0;
        fixed (byte* pOld = input)
        fixed (byte* pPatch = diff)
        {
            if (Native.dq_bspatch_apply(pOld, input.Length, pPatch, diff.Length, null, 0, &newSize) != 0)
            {
                throw new InvalidOperationException(Native.LastError());          // "Corrupt patch"
            }

            var result = new byte[newSize];
            fixed (byte* pOut = result)
            {
                if (Native.dq_bspatch_apply(pOld, input.Length, pPatch, diff.Length, pOut, newSize, &newSize) != 0)
                {
                    throw new InvalidOperationException(Native.LastError());
                }
            }

            output.Write(result, 0, result.Length);
        }
    }
}

/// <summary>Batched <c>Search</c> (Diff.cs:267-298) for many scan positions of newData at once.</summary>
public static class HipMatchSearch
{
    /// <param name="I">suffix array of oldData (n entries, or n + 1 with Diff.Create's zeroed sentinel slot)</param>
    /// <param name="scans">positions in newData; pos[q], len[q] = what Search returns for newData[scans[q]..]</param>
    public static unsafe void Search(ReadOnlySpan<int> I, ReadOnlySpan<byte> oldData, ReadOnlySpan<byte> newData,
                                     ReadOnlySpan<long> scans, Span<int> pos, Span<int> len, int device = -1)
    {
        if (pos.Length != scans.Length || len.Length != scans.Length)
        {
            throw new ArgumentException("pos and len take one entry per scan position");
        }

        if (I.Length != oldData.Length && I.Length != oldData.Length + 1)
        {
            throw new ArgumentException("I must hold one entry per byte of oldData (+ optionally the sentinel slot)");
        }

        fixed (int* pI = I)
        fixed (byte* pOld = oldData)
        fixed (byte* pNew = newData)
        fixed (long* pScans = scans)
        fixed (int* pPos = pos)
        fixed (int* pLen = len)
        {
            Native.Check(Native.dq_bsdiff_search_i32(pOld, oldData.Length, pI, pNew, newData.Length, pScans, 0, scans.Length, 0,
                                                     pPos, pLen, device), nameof(Native.dq_bsdiff_search_i32));
        }
    }
}

/// <summary>
/// One old file, many new files: the suffix array of the old file (Diff.cs:89-90) is built once and stays on the
/// device.  The old file's memory is pinned for the life of the index (the scan loop walks it).
/// </summary>
public sealed unsafe class HipDiffIndex : IDisposable
{
    private readonly byte[] _old;
    private GCHandle _pin;
    private IntPtr _index;

    public HipDiffIndex(byte[] oldData, int device = -1)
    {
        _old = oldData ?? throw new ArgumentNullException(nameof(oldData));
        _pin = GCHandle.Alloc(_old, GCHandleType.Pinned);
        IntPtr ix;
        int rc = Native.dq_bsdiff_index_create((byte*)_pin.AddrOfPinnedObject(), _old.Length, null, null, device, &ix);
        if (rc != 0)
        {
            _pin.Free();
            Native.Check(rc, nameof(Native.dq_bsdiff_index_create));
        }

        _index = ix;
    }

    private HipDiffIndex(byte[] oldData, IntPtr index)
    {
        _old = oldData;
        _pin = GCHandle.Alloc(_old, GCHandleType.Pinned);      // (a second pin of the same array: each copy frees its own)
        _index = index;
    }

    /// <summary>
    /// One more copy of this index on <paramref name="device"/> (dq_bsdiff_index_clone): the suffix array travels device
    /// to device over xGMI instead of being sorted again -- the exchange step of the many-files path, without a
    /// collective library in the process.  Clone to each device of the node from a thread of its own: xGMI is point to
    /// point, so the copies use different links.
    /// </summary>
    public HipDiffIndex Clone(int device)
    {
        if (_index == IntPtr.Zero)
        {
            throw new ObjectDisposedException(nameof(HipDiffIndex));
        }

        IntPtr ix;
        Native.Check(Native.dq_bsdiff_index_clone(_index, device, &ix), nameof(Native.dq_bsdiff_index_clone));
        return new HipDiffIndex(_old, ix);
    }

    /// <summary>The patch <c>Diff.Create(oldData, newData, ...)</c> writes.</summary>
    public byte[] Create(ReadOnlySpan<byte> newData)
    {
        if (_index == IntPtr.Zero)
        {
            throw new ObjectDisposedException(nameof(HipDiffIndex));
        }

        long cap = Native.dq_bsdiff_patch_bound(_old.Length, newData.Length);
        var patch = new byte[cap];
        long len = 0;
        fixed (byte* pNew = newData)
        fixed (byte* pPatch = patch)
        {
            Native.Check(Native.dq_bsdiff_index_diff(_index, pNew, newData.Length, pPatch, cap, &len),
                         nameof(Native.dq_bsdiff_index_diff));
        }

        Array.Resize(ref patch, checked((int)len));
        return patch;
    }

    public void Dispose()
    {
        if (_index != IntPtr.Zero)
        {
            Native.dq_bsdiff_index_free(_index);
            _index = IntPtr.Zero;
            _pin.Free();
        }

        GC.SuppressFinalize(this);
    }

    ~HipDiffIndex() => Dispose();
}
