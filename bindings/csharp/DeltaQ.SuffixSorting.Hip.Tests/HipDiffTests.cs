// HipDiffTests.cs -- the reference's delta tests (test/DeltaQ.BsDiff.Tests/BsDiffTests.cs:30-78) with the HIP pieces in
// place of the managed ones, in every combination that has to interoperate:
//   Diff.Create with HipSuffixSort injected (the literal drop-in, Diff.cs:27,89-90)  -> Patch.Apply
//   HipDiff.Create (everything native)                                               -> Patch.Apply (managed reader)
//   Diff.Create (all managed)                                                        -> HipDiff.Apply (native reader)
// Source only: no dotnet SDK in the build image.  tests/test_gpu_bsdiff.py runs the same round trips through the C ABI.
using DeltaQ.BsDiff;
using DeltaQ.SuffixSorting.Hip;
using DeltaQ.SuffixSorting.LibDivSufSort;
using System;
using System.Collections.Generic;
using System.IO;
using System.Linq;
using Xunit;

namespace DeltaQ.Tests;

public sealed class HipDiffTests
{
    private static byte[] RandomBytes(int size, int seed)
    {
        var bytes = new byte[size];
        new Random(seed).NextBytes(bytes);
        return bytes;
    }

    public static IEnumerable<object[]> Pairs()
    {
        foreach (int size in new[] { 0, 1, 512, 999, 1024, 4096, 1 << 20 })
        {
            byte[] a = RandomBytes(size, 63 * 13 * 63 * 13);
            yield return new object[] { a, RandomBytes(size, 7) };                      // unrelated
            yield return new object[] { a, (byte[])a.Clone() };                         // identical
            if (size >= 512)
            {
                byte[] edited = a.Take(size / 3).Concat(RandomBytes(17, 9)).Concat(a.Skip(size / 2)).ToArray();
                yield return new object[] { a, edited };                               // a deletion and an insertion
            }
        }
    }

    private static byte[] ManagedApply(byte[] oldData, byte[] patch)
    {
        using var output = new MemoryStream();
        Patch.Apply(oldData, patch, output);
        return output.ToArray();
    }

    [Theory]
    [MemberData(nameof(Pairs))]
    public void DiffCreateWithTheHipProviderInjected(byte[] oldData, byte[] newData)
    {
        using var patch = new MemoryStream();
        Diff.Create(oldData, newData, patch, new HipSuffixSort());
        using var managed = new MemoryStream();
        Diff.Create(oldData, newData, managed, new LibDivSufSort());
        Assert.Equal(managed.ToArray(), patch.ToArray());                                // same suffix array -> same patch, byte for byte
        Assert.Equal(newData, ManagedApply(oldData, patch.ToArray()));
    }

    [Theory]
    [MemberData(nameof(Pairs))]
    public void NativeCreateIsReadByTheManagedReaderAndBack(byte[] oldData, byte[] newData)
    {
        using var patch = new MemoryStream();
        HipDiff.Create(oldData, newData, patch);
        Assert.Equal(newData, ManagedApply(oldData, patch.ToArray()));

        using var managed = new MemoryStream();
        Diff.Create(oldData, newData, managed, new LibDivSufSort());
        using var output = new MemoryStream();
        HipDiff.Apply(oldData, managed.ToArray(), output);
        Assert.Equal(newData, output.ToArray());
    }

    [Fact]
    public void OneOldManyNew()
    {
        byte[] oldData = RandomBytes(1 << 20, 3);
        using var index = new HipDiffIndex(oldData);
        for (int k = 0; k < 5; k++)
        {
            byte[] newData = oldData.Take(200_000 * k).Concat(RandomBytes(100 + k, k)).Concat(oldData.Skip(200_000 * k + 50)).ToArray();
            byte[] patch = index.Create(newData);
            Assert.Equal(HipDiff.CreateBytes(oldData, newData, -1, out long len).AsSpan(0, (int)len).ToArray(), patch);
            Assert.Equal(newData, ManagedApply(oldData, patch));
        }
    }

    [Fact]
    public void ACloneOfAnIndexGivesTheSamePatchesAndOutlivesItsSource()
    {
        // dq_bsdiff_index_clone: text + suffix array + prefix table copied device to device (xGMI between the devices of a
        // node); here onto the same device, as tests/test_gpu_bsdiff.py::test_index_clone_gives_the_same_patches does
        byte[] oldData = RandomBytes(1 << 20, 5);
        byte[] newData = oldData.Take(300_000).Concat(RandomBytes(77, 9)).Concat(oldData.Skip(300_040)).ToArray();
        HipDiffIndex copy;
        using (var index = new HipDiffIndex(oldData))
        {
            copy = index.Clone(0);
        }

        using (copy)
        {
            byte[] patch = copy.Create(newData);
            Assert.Equal(HipDiff.CreateBytes(oldData, newData, -1, out long len).AsSpan(0, (int)len).ToArray(), patch);
            Assert.Equal(newData, ManagedApply(oldData, patch));
            long[] info = HipDiff.LastDiffInfo();                 // searches, windows, stop points, host-loop fall-backs, workgroups
            Assert.Equal(5, info.Length);
            Assert.True(info[0] > 0);
            Assert.Equal(0, info[3]);                             // an idle device: the scan ran on it
        }
    }

    [Fact]
    public void CorruptPatchesAreRejectedLikeTheReference()
    {
        byte[] oldData = RandomBytes(4096, 1);
        using var good = new MemoryStream();
        HipDiff.Create(oldData, RandomBytes(4096, 2), good);
        byte[] bad = good.ToArray();
        bad[0] ^= 1;                                                                    // signature
        var ex = Assert.Throws<InvalidOperationException>(() => HipDiff.Apply(oldData, bad, new MemoryStream()));
        Assert.Equal("Corrupt patch", ex.Message);
    }

    [Fact]
    public void BadOutputStreamsThrowLikeDiffCreate()
    {
        Assert.Throws<ArgumentNullException>(() => HipDiff.Create(Array.Empty<byte>(), Array.Empty<byte>(), null!));
        Assert.Throws<ArgumentException>(() => HipDiff.Create(Array.Empty<byte>(), Array.Empty<byte>(), new MemoryStream(Array.Empty<byte>(), false)));
    }

    [Fact]
    public void SearchAgreesWithTheScanLoopsOwn()
    {
        byte[] oldData = RandomBytes(100_000, 5), newData = oldData.Skip(777).Concat(RandomBytes(300, 6)).ToArray();
        using var owner = new HipSuffixSort().Sort(oldData);
        long[] scans = Enumerable.Range(0, 2000).Select(i => (long)i * 37).ToArray();
        var pos = new int[scans.Length];
        var len = new int[scans.Length];
        HipMatchSearch.Search(owner.Memory.Span, oldData, newData, scans, pos, len);
        for (int q = 0; q < scans.Length; q++)
        {
            // the answer is a match: len bytes of old at pos equal new at scan, and it cannot be extended
            int s = (int)scans[q];
            Assert.True(oldData.AsSpan(pos[q], len[q]).SequenceEqual(newData.AsSpan(s, len[q])));
            Assert.True(pos[q] + len[q] == oldData.Length || s + len[q] == newData.Length || oldData[pos[q] + len[q]] != newData[s + len[q]]);
        }
    }
}
